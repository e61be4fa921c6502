#!/usr/bin/env python3
"""Decoder kernel time for every (BG, Z): 25 fixed iterations, all layers, fp16 LLRs resident in HBM, batch
chosen so that every launch carries about the same number of code bits as the headline (4096 x Z=384).
Writes gpurun_out/bench_all_z.json; the per-size landscape cited in DESIGN.md."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
ALL_Z = sorted(a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384)
if os.environ.get("ALLZ_ONLY"):  # a subset of the lifting sizes (A/B sessions)
    ALL_Z = [z for z in ALL_Z if z in {int(x) for x in os.environ["ALLZ_ONLY"].split(",")}]
out = []
for bg in (() if os.environ.get("STOP") else (1, 2)):
    rows, cols, kb = DIMS[bg]
    for Z in ALL_Z:
        B = max(4096, min(262144, (4096 * 384 // Z) // 256 * 256))
        c = pkg.Codec(bg, Z, max_iter=25, early_term=False, llr_dtype=np.float16)
        llr = (torch.randn((B, cols * Z), device="cuda") * 2 + 1.5).half()
        hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
        c.set_timing(True)
        ms = []
        for i in range(4):
            c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            ms.append(c.last_kernel_ms())
        c.close()
        t = min(ms[1:])
        rec = {"bg": bg, "Z": Z, "batch": B, "kernel_ms": t, "info_Gbit_s": B * kb * Z / t / 1e6,
               "edge_updates_per_ns": B * 25 * (316 if bg == 1 else 197) * Z / t / 1e6}
        out.append(rec)
        print("BG%d Z=%3d batch %6d: %.3f ms  %.2f Gbit/s info  %.1f edge-updates/ns" % (bg, Z, B, t, rec["info_Gbit_s"], rec["edge_updates_per_ns"]), flush=True)
        del llr, hard
# STOP=1: the same landscape with the reference's parity-check stop, on valid codewords (the library's own encoder) over QPSK/AWGN at
# the Es/N0 where the stop takes about 10 of 25 iterations on average (bisection on 256 codewords per size) -- which kernel a
# size runs there is a per-size choice (NRLDPC_Z64I_LIST modes, NRLDPC_Z64P_NOT_ET): this is what those choices add up to
if os.environ.get("STOP"):
    out = []
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for bg in (1, 2):
        rows, cols, kb = DIMS[bg]
        for Z in ALL_Z:
            B = max(4096, min(262144, (4096 * 384 // Z) // 256 * 256))
            nb = min(B, max(256, 98304 // Z // 256 * 256))
            B = B // nb * nb
            c = pkg.Codec(bg, Z, max_iter=25, early_term=True, llr_dtype=np.float16)
            info = np.random.default_rng(Z).integers(0, 2, (nb, kb * Z), dtype=np.uint8)
            sgn = 1.0 - 2.0 * torch.from_numpy(c.encode(info).astype(np.float32)).cuda()
            noise = torch.randn(sgn.shape, device="cuda", generator=g)
            def llr_at(esn0):
                mu = 2.0 * 10.0 ** (esn0 / 10.0)
                x = sgn * mu + (2 * mu) ** 0.5 * noise
                x[:, : 2 * Z] = 0
                return x.half().contiguous()
            hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
            iters = torch.zeros(B, device="cuda", dtype=torch.int32)
            lo, hi = -4.0, 4.0
            for _ in range(8):
                mid = 0.5 * (lo + hi)
                x = llr_at(mid)
                c.decode_dev(x.data_ptr(), nb, hard.data_ptr(), iters.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if iters[:nb].float().mean().item() > 10: lo = mid
                else: hi = mid
            esn0 = 0.5 * (lo + hi)
            llr = llr_at(esn0).repeat(B // nb, 1).contiguous()
            c.set_timing(True)
            ms = []
            for i in range(4):
                c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), iters.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                ms.append(c.last_kernel_ms())
            torch.cuda.synchronize()
            c.close()
            t = min(ms[1:])
            rec = {"bg": bg, "Z": Z, "batch": B, "EsN0_dB": esn0, "mean_iters": iters.float().mean().item(), "kernel_ms": t,
                   "info_Gbit_s": B * kb * Z / t / 1e6}
            out.append(rec)
            print("BG%d Z=%3d batch %6d @ %.2f dB: %.3f ms  mean %.2f iterations  %.2f Gbit/s info" % (bg, Z, B, esn0, t, rec["mean_iters"], rec["info_Gbit_s"]), flush=True)
            del llr, hard, sgn, noise
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_all_z_stop%s.json" % os.environ.get("OUT_SUFFIX", "")), "w"), indent=1)
    sys.exit(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_all_z%s.json" % os.environ.get("OUT_SUFFIX", "")), "w"), indent=1)
