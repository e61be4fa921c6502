#!/usr/bin/env python3
"""Generate tests/golden/nmsq_golden.npz: seeded inputs and the ORACLE's outputs for them.

The reference repository holds no decoder vectors and cannot run here (MATLAB + Communications
Toolbox), so these fixtures pin the build's own algorithm (regression + GPU parity), not the
reference's arithmetic.  Encoder outputs in the fixture additionally satisfy H*c = 0.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

CASES = [  # name, bg, Z, n_layers, iters, early_term, alpha, scale, EsN0 dB, batch
    ("cfg1_bg2_z20", 2, 20, 12, 10, 1, 0.75, 8, 1.0, 6),
    ("bg1_z384_r13", 1, 384, 0, 8, 0, 0.75, 8, -0.8, 2),
    ("bg1_z384_r89_et", 1, 384, 5, 25, 1, 0.75, 8, 6.3, 3),
    ("bg2_z384_et", 2, 384, 22, 12, 1, 0.6875, 16, 0.5, 2),
    ("bg1_z2", 1, 2, 0, 10, 1, 0.75, 4, 3.0, 9),
    ("bg2_z208", 2, 208, 21, 8, 1, 0.75, 8, 1.0, 2),
]


def main():
    out = {}
    rng = np.random.default_rng(20260929)
    for name, bg, Z, nl, it, et, alpha, scale, esn0, B in CASES:
        rows, cols, kb = O.BG_DIMS[bg]
        info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
        cw = O.encode(bg, Z, info)
        assert all(O.syndrome_weight(bg, Z, cw[b]) == 0 for b in range(B))
        mu = 2 * 10 ** (esn0 / 10)
        llr = ((1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * rng.standard_normal(cw.shape))
        llr[:, : 2 * Z] = 0
        act = (kb + (nl or rows)) * Z
        llr[:, act:] = 0
        llr = llr.astype(np.float16)
        h, iters, app = O.decode_nmsq(bg, Z, llr.astype(np.float64), it, n_layers=nl, early_term=bool(et),
                                      alpha=alpha, scale=scale, want_app=True)
        out[name + "/cfg"] = np.array([bg, Z, nl, it, et, alpha, scale], np.float64)
        out[name + "/info"] = info
        out[name + "/cw_packed"] = np.packbits(cw, axis=1)
        out[name + "/llr"] = llr
        out[name + "/hard_packed"] = np.packbits(h, axis=1)
        out[name + "/iters"] = iters
        out[name + "/app_f16"] = app.astype(np.float16)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nmsq_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
