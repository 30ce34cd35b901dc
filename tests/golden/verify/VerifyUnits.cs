// VerifyUnits.cs -- pins tests/golden/unit_vectors.npz (here: its text twin unit_vectors.txt) against the UNMODIFIED reference's unit
// entry points: the encoder-side second statements of the path's arithmetic (SURVEY.md 8(c), in-source redundancies 1-3), which the
// reference exposes as static methods:
//   MobiEncoder.IDCT64 / IDCT16 (Encoder/MobiEncoder.cs:1012, 1180)    MobiEncoder.DCT64 / DCT16 (:962, 1146)
//   FrameUtil.GetPBlock (Utils/FrameUtil.cs:96-143)                    MacroBlock.GetCompvals8x8 / GetCompvals4x4 (Encoder/MacroBlock.cs:630, 1184)
//   MacroBlock.PredictIntraPlane16x16 / 8x8 / 4x4 (:1477, 1630, 1716)
// (two of them private: called through reflection).  The decoder's own PredictIntra / CopyBlock / IDCT variants are private instance
// code behind DecodeFrame(): VerifyGolden.cs covers those through whole-frame decodes.  Cannot run in this repository's containers (no
// mono / .NET); anywhere it can:
//   (build LibMobiclip.dll from the reference's LibMobiclip.csproj, or compile its sources)   mcs -r:LibMobiclip.dll -out:VerifyUnits.exe VerifyUnits.cs
//   mono VerifyUnits.exe <repo>/tests/golden        exit code 0 = every vector is the reference's own output
// Vectors whose expected result is "the reference throws" (rc != 0) must throw here too.  For intra predictors the stored result is the
// DECODER statement's; the encoder's copy fetches whole neighbour blocks and throws at the plane's first rows / columns where the decoder
// does not (tests/golden/make_unit_vectors.py): such cases are counted as "encoder throws", not as differences.
using System;
using System.IO;
using System.Linq;
using System.Reflection;
using LibMobiclip.Codec.Mobiclip.Encoder;
using LibMobiclip.Utils;

public static class VerifyUnits
{
    static byte[] Hex(string s) { return Enumerable.Range(0, s.Length / 2).Select(i => Convert.ToByte(s.Substring(2 * i, 2), 16)).ToArray(); }
    static int[] Csv(string s) { return s.Split(',').Select(int.Parse).ToArray(); }
    static bool Same(byte[] a, byte[] b, int n) { for (int i = 0; i < n; i++) if (a[i] != b[i]) return false; return true; }
    static object Priv(string name, params object[] args)
    {
        MethodInfo m = typeof(MacroBlock).GetMethod(name, BindingFlags.NonPublic | BindingFlags.Public | BindingFlags.Static);
        try { return m.Invoke(null, args); } catch (TargetInvocationException e) { throw e.InnerException; }
    }

    public static int Main(string[] args)
    {
        string path = args.Length == 1 ? Path.Combine(args[0], "unit_vectors.txt") : null;
        if (path == null || !File.Exists(path)) { Console.Error.WriteLine("usage: VerifyUnits <dir with unit_vectors.txt (tests/golden)>"); return 2; }
        int bad = 0, n = 0, encThrows = 0, stride = 256;
        byte[] copySrc = null, plane = null;
        int copyOffset = 0;
        foreach (string line in File.ReadAllLines(path))
        {
            string[] t = line.Split(new[] { ' ' }, StringSplitOptions.RemoveEmptyEntries);
            if (t.Length == 0 || t[0].StartsWith("#")) continue;
            bool ok = true;
            switch (t[0])
            {
                case "stride": stride = int.Parse(t[1]); continue;
                case "copysrc": copyOffset = int.Parse(t[1]); copySrc = Hex(t[2]); continue;
                case "intraplane": plane = Hex(t[1]); continue;
                case "idct8":
                case "idct4":
                {
                    int rc = int.Parse(t[1]); int[] c = Csv(t[2]); byte[] p = Hex(t[3]), want = Hex(t[4]);
                    try { byte[] o = t[0] == "idct8" ? MobiEncoder.IDCT64(c, p) : MobiEncoder.IDCT16(c, p); ok = rc == 0 && Same(o, want, want.Length); }
                    catch (Exception) { ok = rc != 0; }   // the clamp table's domain (MobiConst.cs:587): the reference throws
                    break;
                }
                case "dct8":
                case "dct4":
                {
                    int[] o = t[0] == "dct8" ? MobiEncoder.DCT64(Csv(t[1])) : MobiEncoder.DCT16(Csv(t[1]));
                    ok = o.SequenceEqual(Csv(t[2]));
                    break;
                }
                case "copy":
                {
                    uint w = uint.Parse(t[1]), h = uint.Parse(t[2]);
                    byte[] o = FrameUtil.GetPBlock(copySrc, int.Parse(t[3]), int.Parse(t[4]), w, h, copyOffset, stride), want = Hex(t[5]);
                    ok = Same(o, want, want.Length);
                    break;
                }
                case "plane":
                {
                    int size = int.Parse(t[1]), x = int.Parse(t[2]), y = int.Parse(t[3]), param = int.Parse(t[4]);
                    byte[] d = (byte[])plane.Clone(), want = Hex(t[5]);
                    byte[] o = size == 16 ? MacroBlock.PredictIntraPlane16x16(d, y * stride + x, stride, param)
                             : size == 8 ? MacroBlock.PredictIntraPlane8x8(d, y * stride + x, stride, param)
                             : (byte[])Priv("PredictIntraPlane4x4", d, y * stride + x, stride, param);
                    ok = Same(o, want, want.Length);
                    break;
                }
                case "intra":
                {
                    int m = int.Parse(t[1]), x = int.Parse(t[2]), y = int.Parse(t[3]), uv = int.Parse(t[4]), rc = int.Parse(t[5]);
                    if (rc != 0) continue;                         // the decoder throws there (negative offsets): nothing stored
                    byte[] want = Hex(t[6]);
                    int eoff = (uv != 0 && x >= stride / 2) ? stride / 2 : 0;   // the V plane is Offset = Stride / 2, X counted from there
                    try
                    {
                        byte[] d = (byte[])plane.Clone();
                        byte[] o = m < 10 ? MacroBlock.GetCompvals8x8(m, d, x - eoff, y, stride, eoff) : (byte[])Priv("GetCompvals4x4", m, d, x - eoff, y, stride, eoff);
                        ok = Same(o, want, want.Length);
                    }
                    catch (Exception) { encThrows++; continue; }
                    break;
                }
                default: continue;
            }
            n++;
            if (!ok) { bad++; Console.WriteLine("DIFFERENT: " + line.Substring(0, Math.Min(60, line.Length))); }
        }
        Console.WriteLine("{0} vectors checked, {1} different, {2} intra cases where only the decoder statement runs: {3}", n, bad, encThrows,
                          bad == 0 ? "the unit vectors are the reference's output" : "NOT pinned");
        return bad == 0 ? 0 : 1;
    }
}
