#!/usr/bin/env python3
"""BLER of the GPU decoder (layered NMS-Q, rate-dependent alpha) next to the reference-semantics
flooding sum-product oracle (the stand-in for comm.LDPCDecoder, NRLDPCDecoder.m:120) on identical
noise realisations.  Run on the GPU box; writes gpurun_out/bler_compare.json.
Core-level comparison (code blocks through the decoder core), QPSK/AWGN LLRs, rv0."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

pkg = importlib.import_module("ldpc-3gpp-matlab_amd")

CASES = [  # name, bg, Z, K' (payload+CRC), E, n_layers, iterations, SNR list, blocks
    ("cfg1 BG2 A=100 R=1/3 10it", 2, 20, 116, 300, 12, 10, [0.0, 1.0, 2.0, 3.0, 4.0], 4000),
    ("headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.6, -1.4, -1.2, -1.0, -0.8], 1024),
    ("BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.4, -1.2, -1.0, -0.8, -0.6], 1024),
    ("BG1 Z=384 R=8/9 25it", 1, 384, 8448, 9478, 5, 25, [5.6, 6.0, 6.4, 6.8], 1024),
]


def main():
    out = []
    rng = np.random.default_rng(2026)
    for name, bg, Z, Kp, E, nl, iters, snrs, nblk in CASES:
        rows, cols, kb = O.BG_DIMS[bg]
        K = kb * Z
        alpha = pkg.default_alpha(bg, nl)
        codec = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, alpha=alpha, llr_dtype=np.float32)
        info = rng.integers(0, 2, (nblk, K), dtype=np.uint8)
        info[:, Kp:] = 0
        cw = codec.encode(info)
        noise = rng.standard_normal(cw.shape)
        for snr in snrs:
            mu = 2 * 10 ** (snr / 10)
            llr = (1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * noise
            llr[:, : 2 * Z] = 0
            llr[:, 2 * Z + E + (K - Kp):] = 0     # E transmitted non-filler bits from k0 = 0
            llr[:, Kp:K] = np.inf                  # fillers (NRLDPCDecoder.m:264)
            t0 = time.time()
            hg, itg = codec.decode(llr.astype(np.float32), want_iters=True)
            tg = time.time() - t0
            t0 = time.time()
            hb, itb = O.decode_bp_flood(bg, Z, llr, iters, n_layers=nl)
            tb = time.time() - t0
            hb50, itb50 = O.decode_bp_flood(bg, Z, llr, 50, n_layers=nl)
            rec = {"case": name, "EsN0_dB": snr, "blocks": nblk, "alpha": alpha,
                   "bler_gpu_nmsq": float((hg[:, :Kp] != info[:, :Kp]).any(1).mean()),
                   "bler_bp_flood": float((hb[:, :Kp] != info[:, :Kp]).any(1).mean()),
                   "bler_bp_flood_50it": float((hb50[:, :Kp] != info[:, :Kp]).any(1).mean()),
                   "mean_iters_gpu": float(itg.mean()), "mean_iters_bp": float(itb.mean()),
                   "t_gpu_s": tg, "t_bp_cpu_s": tb}
            print(rec, flush=True)
            out.append(rec)
        codec.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bler_compare.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
