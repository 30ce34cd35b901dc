// VerifyGolden.cs -- pins the committed golden fixtures against the UNMODIFIED reference decoder.
//
// The reference (Gericom/MobiclipDecoder) ships no test vectors, and C# cannot run in the build container of this repository, so
// tests/golden/*.bin + golden.json hold streams from this repository's seeded generator and the SHA-256 of every plane as the CPU
// oracle (oracle/mobi_oracle.c) decodes them: "parity unpinned" (DESIGN.md (c)).  This program is the one-command job that turns
// that into "pinned" anywhere .NET or mono exists: it decodes every fixture with LibMobiclip.Codec.Mobiclip.MobiclipDecoder exactly as
// the reference's own callers do (MobiConverter/Program.cs:57-71: d.Data = frame; d.Offset = o; d.DecodeFrame(); MobiConverter/
// Program.cs:243-250 reads d.Offset afterwards) and compares the SHA-256 of d.Y[0] / d.UV[0], d.Offset and d.Quantizer with the manifest.
//
//   tests/golden/verify/build.sh <reference checkout>     builds and runs this program and VerifyUnits with mcs / mono or with dotnet; by hand:
//   mcs -unsafe -r:System.Drawing.dll -out:VerifyGolden.exe VerifyGolden.cs <reference>/LibMobiclip/Codec/Mobiclip/MobiclipDecoder.cs \
//       <reference>/LibMobiclip/Codec/Mobiclip/MobiConst.cs <reference>/LibMobiclip/Utils/IOUtil.cs
//   mono VerifyGolden.exe <repo>/tests/golden
//
// DecodeFrame() also builds a System.Drawing.Bitmap (MobiclipDecoder.cs:260-323).  Where libgdiplus is missing that throws inside the
// decoder's own try / catch (:99, :325-328), after the planes are complete: the planes, Offset and Quantizer compared here are unaffected.
// Exit code 0 = every frame of every fixture identical; 1 = differences (listed); 2 = usage / missing files.
using System;
using System.Globalization;
using System.IO;
using System.Security.Cryptography;
using LibMobiclip.Codec.Mobiclip;

public static class VerifyGolden
{
    static string Sha(byte[] b)
    {
        using (SHA256 h = SHA256.Create())
            return BitConverter.ToString(h.ComputeHash(b)).Replace("-", "").ToLowerInvariant();
    }

    public static int Main(string[] args)
    {
        if (args.Length != 1 || !File.Exists(Path.Combine(args[0], "golden_manifest.txt")))
        {
            Console.Error.WriteLine("usage: VerifyGolden <dir with golden_manifest.txt and the .bin fixtures (tests/golden)>");
            return 2;
        }
        string dir = args[0];
        int bad = 0, frames = 0, badInCase = 0;
        MobiclipDecoder d = null;
        byte[] data = null;
        string name = null, covers = null;
        // golden_manifest.txt (written by tests/golden/make_golden.py next to golden.json):
        //   case <name> <width> <height> <version 1=ModsDS 2=Moflex3DS> <n_frames>
        //   covers <what the fixture exercises, from the oracle's coverage counters: what a run without differences pins>
        //   frame <start offset> <end offset> <sha256 of Y[0]> <sha256 of UV[0]> <Offset after DecodeFrame> <Quantizer>
        //   reject <the same fields>: DecodeFrame() returns null for this frame; the hashes are those of the partial picture it leaves
        string[] lines = File.ReadAllLines(Path.Combine(dir, "golden_manifest.txt"));
        for (int li = 0; li <= lines.Length; li++)
        {
            string line = li < lines.Length ? lines[li] : "case";   // (a last, empty "case" closes the last fixture)
            string[] t = line.Split(new[] { ' ' }, StringSplitOptions.RemoveEmptyEntries);
            if (t.Length == 0 || t[0].StartsWith("#")) continue;
            if (t[0] == "case")
            {
                if (name != null)
                    Console.WriteLine("{0} {1}: {2}", badInCase == 0 ? "pins" : "DOES NOT PIN", name, covers ?? "");
                badInCase = 0;
                covers = null;
                if (t.Length < 6) continue;
                name = t[1];
                data = File.ReadAllBytes(Path.Combine(dir, name + ".bin"));
                d = new MobiclipDecoder(uint.Parse(t[2]), uint.Parse(t[3]),
                                        int.Parse(t[4]) == 1 ? MobiclipDecoder.MobiclipVersion.ModsDS : MobiclipDecoder.MobiclipVersion.Moflex3DS);
            }
            else if (t[0] == "covers")
            {
                covers = line.Substring(line.IndexOf(' ') + 1);
            }
            else if (t[0] == "frame" || t[0] == "reject")
            {
                // "reject": a frame the reference returns null for (the exception is swallowed, MobiclipDecoder.cs:325-328).  The hashes are those of
                // the PARTIAL picture it keeps in Y[0] / UV[0] -- they pin where the throw happened -- and Offset is where the reader stood.
                int start = int.Parse(t[1], CultureInfo.InvariantCulture), end = int.Parse(t[2], CultureInfo.InvariantCulture);
                byte[] frame = new byte[end];                 // Data = the stream up to the end of this frame, Offset = where it starts
                Array.Copy(data, frame, end);                 // (what tests/golden/make_golden.py handed the oracle)
                d.Data = frame;
                d.Offset = start;
                object bitmap = d.DecodeFrame();
                string y = Sha(d.Y[0]), uv = Sha(d.UV[0]);
                bool ok = y == t[3] && uv == t[4] && d.Offset == int.Parse(t[5]) && d.Quantizer == uint.Parse(t[6]);
                if (t[0] == "reject" && bitmap != null) ok = false;   // (the converse is not checked: without libgdiplus every frame returns null, see above)
                frames++;
                if (!ok)
                {
                    bad++;
                    badInCase++;
                    Console.WriteLine("DIFFERENT {0} frame at {1}: Y {2} UV {3} Offset {4} (want {5}) Quantizer {6} (want {7})", name, start,
                                      y == t[3] ? "ok" : y, uv == t[4] ? "ok" : uv, d.Offset, t[5], d.Quantizer, t[6]);
                }
            }
        }
        Console.WriteLine("{0} frames, {1} different: {2}", frames, bad, bad == 0 ? "the golden fixtures are the reference's output (parity pinned)" : "NOT pinned");
        return bad == 0 ? 0 : 1;
    }
}
