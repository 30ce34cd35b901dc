"""The streams of the coverage suite: tests/test_gpu_parity.py::test_coverage_suite_streams decodes every one of them through
the HIP path against the oracle (`-m gpu`), and tests/test_coverage.py (CPU) asserts that their union reaches every part of the
syntax the oracle counts (oracle/mobi_oracle.h, MOBI_COV_*).  So "covered" means covered by a stream the GPU suite really runs.
(config, seed offset, generator overrides)"""
COVERAGE_SUITE = [
    # the three BASELINE configurations with the SURVEY 8(d) mix
    ("A", 0, dict(n_frames=9)),
    ("B", 0, dict(n_frames=9)),
    ("C", 0, dict(n_frames=9)),
    # rich: deep trees, every reference slot, both VLC tables, escapes, quantiser deltas, I-frames in between
    ("A", 77, dict(n_frames=10, pm_intra=150, pm_deep=150, pm_multiref=300, qdelta_prob=300, table1_prob=500, escape_prob=100, iframe_interval=6)),
    ("B", 77, dict(n_frames=10, pm_intra=150, pm_deep=150, pm_multiref=300, qdelta_prob=300, table1_prob=500, escape_prob=100, iframe_interval=6)),
    # almost nothing but deep partition trees (every shape down to 2x2, both versions)
    ("A", 401, dict(n_frames=8, pm_deep=700, pm_split1=200, pm_skip=50, pm_intra=20, pm_multiref=400, cbp_prob=100)),
    ("B", 402, dict(n_frames=8, pm_deep=700, pm_split1=200, pm_skip=50, pm_intra=20, pm_multiref=400, cbp_prob=100)),
    # intra-heavy: every directional mode in both block sizes, the three plane predictors
    ("A", 403, dict(n_frames=6, pm_intra=600, intra_sub_prob=800, plane_prob=600, t8_prob=400)),
    ("B", 404, dict(n_frames=6, pm_intra=600, intra_sub_prob=800, plane_prob=600, t8_prob=400)),
    # residual variety: single-coefficient and dense blocks (all four 8x8 and both 4x4 transform classes), escapes of every kind
    ("A", 405, dict(n_frames=6, cbp_prob=800, max_coefs=1, scan_span=1, t8_prob=500)),
    ("A", 406, dict(n_frames=6, cbp_prob=700, max_coefs=2, scan_span=3, t8_prob=700, escape_prob=400)),
    ("B", 407, dict(n_frames=6, cbp_prob=700, dense_prob=500, escape_prob=300, table1_prob=500, t8_prob=600)),
]


def suite_params():
    from mobiclipdecoder_amd import default_params
    from mobiclipdecoder_amd.streamgen import BASE_SEED
    return [default_params(cfg, BASE_SEED + seed, **kw) for cfg, seed, kw in COVERAGE_SUITE]
