#!/usr/bin/env python3
"""Looks for an error floor of the decoder (offset min-sum, int8 messages) beyond the waterfall: millions of codewords per
point, generated and decoded on the GPU (encode_dev, torch noise, decode_dev with the library's default rule and the
parity-check stop), block errors counted on the device.  Writes gpurun_out/bler_floor.json.
    python tools/bler_floor.py [blocks_per_point]"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
NBLK = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
CASES = [  # name, bg, Z, K', E, n_layers, iterations, Es/N0 points
    ("headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.2, -1.0, -0.8, -0.5]),
    ("BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.0, -0.8, -0.5]),
    ("BG2 Z=384 R=2/3 25it", 2, 384, 3840, 5736, 7, 25, [3.6, 4.0, 4.5]),
    ("BG1 Z=384 R=8/9 25it", 1, 384, 8448, 9478, 5, 25, [6.6, 7.0, 7.5]),
    ("cfg1 BG2 A=100 (Z=20) R=1/3 10it", 2, 20, 116, 300, 12, 10, [3.0, 4.0, 5.0, 6.0]),
]
out = []
for name, bg, Z, Kp, E, nl, iters, snrs in CASES:
    rows, cols, kb = DIMS[bg]
    K, N = kb * Z, cols * Z
    B = 8192 if Z >= 100 else 65536
    c = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=True, llr_dtype=np.float16)
    g = torch.Generator(device="cuda"); g.manual_seed(2026)
    s = torch.cuda.current_stream().cuda_stream
    info = torch.empty((B, K), device="cuda", dtype=torch.uint8); cw = torch.empty((B, N), device="cuda", dtype=torch.uint8)
    hard = torch.empty((B, K), device="cuda", dtype=torch.uint8); it = torch.empty(B, device="cuda", dtype=torch.int32)
    for snr in snrs:
        mu = 2 * 10 ** (snr / 10)
        errs = blocks = 0; its = 0.0
        while blocks < NBLK and errs < 2000:
            info.random_(0, 2, generator=g); info[:, Kp:] = 0
            c.encode_dev(info.data_ptr(), B, cw.data_ptr(), s)
            llr = (1 - 2 * cw.float()) * mu + (2 * mu) ** 0.5 * torch.randn(cw.shape, generator=g, device="cuda")
            llr[:, :2 * Z] = 0; llr[:, 2 * Z + E + (K - Kp):] = 0; llr[:, Kp:K] = float("inf")
            llr = llr.half().contiguous()
            c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), it.data_ptr(), None, s)
            errs += int((hard[:, :Kp] != info[:, :Kp]).any(1).sum()); blocks += B; its += float(it.float().sum())
        rec = {"case": name, "EsN0_dB": snr, "blocks": blocks, "block_errors": errs, "bler": errs / blocks, "mean_iters": its / blocks,
               "alpha": c.alpha, "beta_llr": c.beta}
        print(rec, flush=True); out.append(rec)
    c.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bler_floor.json"), "w"), indent=1)
