#!/bin/sh
# Pin this repository's golden fixtures and unit vectors against the REAL reference, in one command:
#
#     tests/golden/verify/build.sh /path/to/Gericom/MobiclipDecoder
#
# Needs mono (mcs + mono) or the .NET SDK (dotnet); neither exists in this repository's build containers, which is why parity is "unpinned"
# (DESIGN.md (c)).  The reference checkout is used UNMODIFIED: its sources are compiled where they lie, nothing is copied or patched.
#   VerifyGolden  = MobiclipDecoder.cs + MobiConst.cs + IOUtil.cs driven like MobiConverter/Program.cs:57-71 drives them, against
#                   tests/golden/golden_manifest.txt (SHA-256 of Y[0] / UV[0], Offset, Quantizer of every frame of every fixture)
#   VerifyUnits   = all of LibMobiclip (the encoder-side copies of the transforms, predictors and CopyBlock) against tests/golden/unit_vectors.txt
# Exit code 0 = both programs printed "parity pinned" / "the unit vectors are the reference's output".  Paste their last lines (and the
# "pins:" lines) into DESIGN.md (c) and delete the words "parity unpinned" there and in oracle/mobi_oracle.c's header.
set -e
REF=${1:?usage: build.sh <checkout of Gericom/MobiclipDecoder>}
HERE=$(cd "$(dirname "$0")" && pwd)
GOLDEN=$(dirname "$HERE")
OUT=${TMPDIR:-/tmp}/mobiclip_verify
mkdir -p "$OUT"
DEC="$REF/LibMobiclip/Codec/Mobiclip/MobiclipDecoder.cs $REF/LibMobiclip/Codec/Mobiclip/MobiConst.cs $REF/LibMobiclip/Utils/IOUtil.cs"
for f in $DEC; do [ -f "$f" ] || { echo "not a checkout of the reference: $f is missing" >&2; exit 2; }; done
if command -v mcs >/dev/null 2>&1 && command -v mono >/dev/null 2>&1; then
  mcs -nologo -unsafe -r:System.Drawing.dll -r:System.Core.dll -out:"$OUT/VerifyGolden.exe" "$HERE/VerifyGolden.cs" $DEC
  mcs -nologo -unsafe -r:System.Drawing.dll -r:System.Core.dll -out:"$OUT/VerifyUnits.exe" "$HERE/VerifyUnits.cs" $(find "$REF/LibMobiclip" -name '*.cs' ! -path '*/obj/*' ! -path '*/Properties/*')
  mono "$OUT/VerifyGolden.exe" "$GOLDEN"
  mono "$OUT/VerifyUnits.exe" "$GOLDEN"
elif command -v dotnet >/dev/null 2>&1; then
  # two throw-away projects that LINK the reference's sources (System.Drawing.Common: the decoder builds a Bitmap at the end of DecodeFrame();
  # if it cannot, that throws inside the decoder's own try / catch after the planes are complete -- what is compared here is unaffected)
  for P in VerifyGolden VerifyUnits; do
    mkdir -p "$OUT/$P"
    if [ $P = VerifyGolden ]; then SRC=$(for f in $DEC; do printf '<Compile Include="%s" />' "$f"; done)
    else SRC="<Compile Include=\"$REF/LibMobiclip/**/*.cs\" Exclude=\"$REF/LibMobiclip/obj/**;$REF/LibMobiclip/Properties/**\" />"; fi
    cat > "$OUT/$P/$P.csproj" <<XML
<Project Sdk="Microsoft.NET.Sdk">
  <PropertyGroup><OutputType>Exe</OutputType><TargetFramework>net8.0</TargetFramework><AllowUnsafeBlocks>true</AllowUnsafeBlocks>
    <EnableDefaultCompileItems>false</EnableDefaultCompileItems><Nullable>disable</Nullable><NoWarn>CA1416;CS0168;CS0219;CS0414;CS0649</NoWarn></PropertyGroup>
  <ItemGroup><PackageReference Include="System.Drawing.Common" Version="8.0.0" /><Compile Include="$HERE/$P.cs" />$SRC</ItemGroup>
</Project>
XML
    dotnet run --project "$OUT/$P/$P.csproj" -c Release -- "$GOLDEN"
  done
else
  echo "neither mono (mcs) nor the .NET SDK (dotnet) is installed" >&2; exit 2
fi
echo "both checks passed: the golden fixtures and the unit vectors are the reference's own output"
