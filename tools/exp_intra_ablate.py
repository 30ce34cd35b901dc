"""Where mobi_recon_intra's TIME goes (GPU box): the profiling twin of the library with parts of the kernel switched off (MOBI_INTRA_DBG bits:
1 no transforms, 2 no steps, 4 no scatter, 8 no stores, 16 no halo loads, 32 no level words) or left after stage n (n << 8, see
tools/exp_istages.sh); time of the intra launch per P-frame step and of an I-frame step.  Wrong pictures on purpose, faults ignored.
One process per setting (the library reads the variable once).  python tools/exp_intra_ablate.py [clips]"""
import _prof  # noqa: F401  (the profiling twin of the library)
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import mobiclipdecoder_amd as m
    from mobiclipdecoder_amd.streamgen import BASE_SEED
    clips, distinct = int(sys.argv[2]), 8
    b = m.MobiclipBatch(clips, 640, 480, 2)
    for i in range(distinct):
        p = m.default_params("B", BASE_SEED + i, n_frames=33)
        data, fo = m.generate_clip(p)
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips):
        b.preload_clone(c, c % distinct)
    b.commit()
    b.replay(0)
    for f in range(1, 9):
        b.replay(f)
    b.sync()
    b.set_kernel_timing(2)
    b.time_begin()
    for f in range(1, 33):
        b.replay(f)
    b.time_end()
    km = b.kernel_ms()
    b.sync()
    b.time_begin()
    for _ in range(3):
        b.replay(0)
    b.time_end()
    ki = b.kernel_ms()
    print("P-frame intra %.4f ms (inter %.4f)   I-frame intra %.3f ms" % (km["intra_ms"] / max(1, km["intra_launches"]), km["inter_ms"] / max(1, km["inter_launches"]),
                                                                        ki["intra_ms"] / max(1, ki["intra_launches"])), flush=True)
    b.close()
else:
    clips = sys.argv[1] if len(sys.argv) > 1 else "8192"
    for d in [0, 1, 2, 3, 4, 8, 16, 32] + [n << 8 for n in range(1, 8)]:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", clips], env=dict(os.environ, MOBI_INTRA_DBG=str(d)), capture_output=True, text=True, timeout=300)
        print("MOBI_INTRA_DBG=%-5d %s" % (d, r.stdout.strip() or r.stderr.strip()[-200:]), flush=True)
