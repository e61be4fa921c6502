#!/usr/bin/env python3
"""Row form against split form for every compile-time-Z size, in ONE GPU session: run once per form with the A/B library
(NRLDPC_BUILD_AB=1 build; NRLDPC_LIB=.../libnrldpc_hip_ab.so NRLDPC_SPLIT=0|1 OUT_SUFFIX=_row|_split).  Per size: 25 fixed
iterations and the parity-check stop on noisy codewords at the waterfall point, all layers, fp16 LLRs resident in HBM.
Writes gpurun_out/bench_forms<OUT_SUFFIX>.json -- the evidence behind z64_split_default (nrldpc_decode_z64.h).
PAIRS=packed: the packed-geometry sizes instead (run with and without NRLDPC_NO_PACKED=1: the evidence behind NRLDPC_Z64P_LIST)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
bld = importlib.import_module("ldpc-3gpp-matlab_amd.build")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
out = []
for bg, Z in (bld.Z64P_PAIRS if os.environ.get("PAIRS") == "packed" else bld.Z64_PAIRS):
    rows, cols, kb = DIMS[bg]
    B = max(4096, (4096 * 384 // Z) // 256 * 256)
    esn0 = -0.5 if bg == 1 else -1.0
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    enc = pkg.Codec(bg, Z, max_iter=25, llr_dtype=np.float16)
    info = torch.randint(0, 2, (B, kb * Z), generator=g, device="cuda", dtype=torch.uint8)
    cw = torch.empty((B, cols * Z), device="cuda", dtype=torch.uint8)
    enc.encode_dev(info.data_ptr(), B, cw.data_ptr(), torch.cuda.current_stream().cuda_stream)
    enc.close()
    mu = 2.0 * 10 ** (esn0 / 10)
    llr = (1 - 2 * cw.float()) * mu + (2 * mu) ** 0.5 * torch.randn((B, cols * Z), generator=g, device="cuda")
    llr[:, : 2 * Z] = 0
    llr = llr.half().contiguous()
    hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
    rec = {"bg": bg, "Z": Z, "batch": B, "EsN0_dB": esn0}
    for et in (0, 1):
        c = pkg.Codec(bg, Z, max_iter=25, early_term=bool(et), llr_dtype=np.float16)
        c.set_timing(True)
        ms = []
        for i in range(9):
            c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            ms.append(c.last_kernel_ms())
        c.close()
        rec["et%d_ms" % et] = float(np.median(ms[2:]))
    rec["bit_errors"] = int((hard != info).sum().item())
    out.append(rec)
    print("BG%d Z=%3d batch %6d: fixed %.3f ms  parity stop %.3f ms  bit errors %d" % (bg, Z, B, rec["et0_ms"], rec["et1_ms"], rec["bit_errors"]), flush=True)
    del llr, hard, cw, info
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_forms%s.json" % os.environ.get("OUT_SUFFIX", "")), "w"), indent=1)
