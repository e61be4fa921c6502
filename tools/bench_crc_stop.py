#!/usr/bin/env python3
"""CRC-aided stop (nrldpc_cfg.early_term = 2) against the parity-check stop (early_term = 1) at the four operating points of
profiles/r02_crc_aided_stop_potential.txt: mean iterations, kernel time, block errors, and undetected-error safety (a block that
stops on its CRC must equal the transmitted one wherever the parity-check decoder gets it right).  4096 code blocks per point with a
real CRC (payload + CRC24A / CRC16 = K' = K bits), identical LLRs for both modes.  Writes gpurun_out/bench_crc_stop.json."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import BG_DIMS
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
CRC24A, CRC16 = (0x1864CFB, 24), (0x11021, 16)
POINTS = [("BG1 Z=384 R=1/3 (46 rows)", 1, 384, 0, 25344, -0.5, CRC24A), ("BG1 Z=384 R=1/3 (46 rows)", 1, 384, 0, 25344, -1.2, CRC24A),
          ("BG2 Z=384 R=1/3 (22 rows)", 2, 384, 22, 11520, -0.5, CRC16), ("BG1 Z=384 R=8/9 (5 rows)", 1, 384, 5, 9478, 7.5, CRC24A)]
B = 4096
out = []
rng = np.random.default_rng(77)
for name, bg, Z, nl, E, esn0, (poly, L) in POINTS:
    rows, cols, kb = BG_DIMS[bg]
    K = kb * Z
    info = rng.integers(0, 2, (256, K), dtype=np.uint8)
    for b in range(256):
        r = orc.crc(poly, L, info[b, : K - L])
        info[b, K - L:] = (r >> np.arange(L - 1, -1, -1)) & 1
    cw = torch.from_numpy(np.tile(orc.encode(bg, Z, info), (B // 256, 1))).cuda()
    truth = torch.from_numpy(np.tile(info, (B // 256, 1))).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(int(1000 * (esn0 + 10)) + bg)
    mu = 2.0 * 10.0 ** (esn0 / 10.0)
    llr = (1.0 - 2.0 * cw.float()) * mu + (2.0 * mu) ** 0.5 * torch.randn(cw.shape, generator=g, device="cuda")
    llr[:, : 2 * Z] = 0
    llr[:, 2 * Z + E:] = 0
    llr = llr.half().contiguous()
    rec = {"point": name, "EsN0_dB": esn0, "n_layers": nl or rows, "blocks": B, "crc_bits": L}
    res = {}
    for mode, crc in (("parity_check_stop", None), ("crc_aided_stop", (poly, L, K))):
        c = pkg.Codec(bg, Z, max_iter=25, n_layers=nl, early_term=True, llr_dtype=np.float16, crc=crc)
        hard = torch.empty((B, K), device="cuda", dtype=torch.uint8)
        it = torch.empty(B, device="cuda", dtype=torch.int32)
        c.set_timing(True)
        ms = []
        for _ in range(9):
            c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            ms.append(c.last_kernel_ms())
        c.close()
        res[mode] = (hard.clone(), it.clone())
        rec[mode] = {"kernel_ms": float(np.median(ms[2:])), "mean_iterations": float(it.float().mean()),
                     "block_errors": int((hard != truth).any(1).sum())}
    h1, i1 = res["parity_check_stop"]; h2, i2 = res["crc_aided_stop"]
    early = i2 < i1
    rec["blocks_stopped_earlier_by_crc"] = int(early.sum())
    rec["of_those_wrong"] = int(((h2 != truth).any(1) & early).sum())  # undetected errors the CRC let through (none expected at 2^-24 / 2^-16)
    rec["iterations_saved"] = rec["parity_check_stop"]["mean_iterations"] - rec["crc_aided_stop"]["mean_iterations"]
    rec["time_ratio_crc_over_parity"] = rec["crc_aided_stop"]["kernel_ms"] / rec["parity_check_stop"]["kernel_ms"]
    out.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_crc_stop.json"), "w"), indent=1)
