#!/usr/bin/env python3
"""Extended differential fuzz of the decoder kernels against the CPU oracle (the test-suite version runs 60
cases; this runs N, default 600, biased towards the compile-time-Z sizes).  python tools/fuzz_decode.py [N] [seed]
SMALL=1: biased towards the packed-geometry sizes (Z <= 80): pipelined builds (every row active, hard output) and the general
kernel (pruned rows, soft output), ragged batches of up to 70 codewords.
REFILL=1: the parity stop on the sizes whose workgroups hold several codewords (Z <= 192), batches of 40 ... 400 codewords decoded by 1 ... 4
workgroups (NRLDPC_REFILL_GRID, read per call), so that every slot is refilled several times; any layer count, any SNR, any cap.  The refill period
is a per-process setting: run once per NRLDPC_REFILL_MASK=0/1/3.
MULTI=1: nrldpc_decode_multi_dev -- each case one call over 2 ... 14 configurations drawn at random ((BG, Z), layer count given / AUTO / all rows,
fp16 or fp32 LLRs, 1 ... 90 codewords each, a bucket now and then large enough for a launch of its own), every configuration against the oracle:
the shared launches' workgroup classes and prefix tables, the routing, the per-handle layer counts."""
import importlib, os, sys
os.environ.setdefault("NRLDPC_TEST_HOOKS", "1")  # NRLDPC_REFILL_GRID is read only under this (nrldpc_decode_z64p.h)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import ALL_Z, BG_DIMS
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
T = importlib.import_module("test_decode_gpu")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
BIG = [z for z in ALL_Z if z >= 52]
SMALL = [z for z in ALL_Z if z <= 80] if os.environ.get("SMALL") else None
from conftest import awgn_llr, rule_kw
for i in range(N):
    if os.environ.get("MULTI"):
        import torch
        et = bool(rng.random() < 0.8); iters = int(rng.integers(1, 11)); snr = float(rng.uniform(-1.0, 6.0))
        work = []
        for _ in range(int(rng.integers(2, 15))):
            bg = int(rng.integers(1, 3)); Z = int(rng.choice(ALL_Z)); rows, cols, kb = BG_DIMS[bg]
            mode = int(rng.integers(0, 3))  # 0 all rows, 1 the count given, 2 AUTO
            nl = rows if mode == 0 else int(rng.integers(4, rows + 1))
            B = int(rng.integers(1, 91)) if Z <= 96 else int(rng.integers(1, 13))
            if rng.random() < 0.04: B = 512 * 384 // Z + 1  # its own launch (NRLDPC_MULTI_Z64_MIN_ROWS)
            dt = [np.float16, np.float32][int(rng.integers(0, 2))]
            info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
            llr = awgn_llr(rng, orc.encode(bg, Z, info), snr, dt, Z, E=(kb + nl - 2) * Z if mode else None)
            c = pkg.Codec(bg, Z, max_iter=iters, n_layers=(0, nl, -1)[mode], early_term=et, llr_dtype=dt)
            d = torch.from_numpy(llr).cuda()
            work.append((c, d, torch.full((B, kb * Z), 7, dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"), bg, Z, nl, llr))
        uniq = work  # (a handle under AUTO may appear once per call: every configuration here has a handle of its own)
        pkg.MultiCall([w[0] for w in uniq], [w[1].data_ptr() for w in uniq], [w[1].shape[0] for w in uniq], [w[2].data_ptr() for w in uniq],
                      [w[3].data_ptr() for w in uniq])(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for c, d, hard, its, bg, Z, nl, llr in uniq:
            a_, b_ = pkg._capi.default_rule(bg, nl)
            ref = orc.decode_nmsq(bg, Z, llr.astype(np.float32).astype(np.float64), iters, n_layers=nl, early_term=et, alpha=a_, beta=b_ * 8)
            assert c.last_layers() == nl, (bg, Z, nl, c.last_layers())
            assert (hard.cpu().numpy() == ref[0]).all() and (its.cpu().numpy() == ref[1]).all(), (bg, Z, nl, llr.shape, llr.dtype, et, iters)
            c.close()
        if i % 20 == 19:
            print(i + 1, "multi cases ok", flush=True)
        continue
    if os.environ.get("AUTO"):  # NRLDPC_LAYERS_AUTO (ABI revision 5): a rate-matched batch (zero tail), the count read off the data, host and pipelined paths
        bg = int(rng.integers(1, 3)); Z = int(rng.choice(ALL_Z)); rows, cols, kb = BG_DIMS[bg]
        nl = int(rng.integers(4, rows + 1)); B = int(rng.integers(1, 9)) if Z > 64 else int(rng.integers(1, 200))
        if rng.random() < 0.08: B = max(B, (9 << 20) // (cols * Z * 2) + 3)  # above 8 MB: the pipelined host path
        dt = [np.float16, np.float32, np.float64][int(rng.integers(0, 3))]
        et = bool(rng.integers(0, 2)); iters = int(rng.integers(1, 13))
        info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
        llr = awgn_llr(rng, orc.encode(bg, Z, info), float(rng.uniform(-2.0, 7.0)), dt, Z, E=(kb + nl - 2) * Z)
        c = pkg.Codec(bg, Z, max_iter=iters, n_layers=-1, early_term=et, llr_dtype=dt)
        h, it = c.decode(llr, want_iters=True)
        used = c.last_layers()
        c.close()
        assert used == nl, (bg, Z, nl, used)
        a_, b_ = pkg._capi.default_rule(bg, nl)
        ref = orc.decode_nmsq(bg, Z, llr.astype(np.float32).astype(np.float64), iters, n_layers=nl, early_term=et, alpha=a_, beta=b_ * 8)
        assert (h == ref[0]).all() and (it == ref[1]).all(), (bg, Z, nl, B, dt, et, iters)
        if i % 50 == 49:
            print(i + 1, "auto cases ok", flush=True)
        continue
    if os.environ.get("REFILL"):
        bg = int(rng.integers(1, 3)); Z = int(rng.choice([z for z in ALL_Z if z <= 192]))
        os.environ["NRLDPC_REFILL_GRID"] = str(int(rng.integers(1, 5)))
        T.run_case(pkg, orc, rng, bg, Z, int(rng.integers(40, 401)) if Z <= 64 else int(rng.integers(8, 60)), float(rng.uniform(-3.0, 6.0)), int(rng.integers(2, 21)),
                   nl=0 if rng.random() < 0.5 else int(rng.integers(4, BG_DIMS[bg][0] + 1)), et=True,
                   dt=[np.float16, np.float32][int(rng.integers(0, 2))], app=False)
        if i % 50 == 49:
            print(i + 1, "refill cases ok", flush=True)
        continue
    if SMALL and rng.random() < 0.85:
        bg = int(rng.integers(1, 3)); Z = int(rng.choice(SMALL))
        T.run_case(pkg, orc, rng, bg, Z, int(rng.integers(1, 1 + max(8, min(70, 600 // Z)))), float(rng.uniform(-3.0, 5.0)), int(rng.integers(1, 13)),
                   nl=0 if rng.random() < 0.5 else int(rng.integers(4, BG_DIMS[bg][0] + 1)), et=bool(rng.integers(0, 2)),
                   dt=[np.float16, np.float32][int(rng.integers(0, 2))], app=bool(rng.random() < 0.3))
        if i % 50 == 49:
            print(i + 1, "cases ok", flush=True)
        continue
    bg = int(rng.integers(1, 3))
    Z = int(rng.choice(BIG)) if rng.random() < 0.75 else int(rng.choice(ALL_Z))
    rows = BG_DIMS[bg][0]
    nl = 0 if rng.random() < 0.4 else int(rng.integers(4, rows + 1))
    if Z == 384 and rng.random() < 0.5:  # the pruned counts with pipelined kernels of their own (NRLDPC_Z64_NL_LIST)
        nl = int(rng.choice([5, 13, 24] if bg == 1 else [32, 22, 17, 12, 9, 7]))
    if rng.random() < 0.2:
        Z = 384
    use_default = rng.random() < 0.3      # cfg.alpha = 0: the library's rule
    T.run_case(pkg, orc, rng, bg, Z, int(rng.integers(1, 8)), float(rng.uniform(-2.0, 8.0)), int(rng.integers(1, 13)),
               nl=nl, et=bool(rng.integers(0, 2)), dt=[np.float16, np.float32][int(rng.integers(0, 2))],
               alpha=None if use_default else float(rng.choice([0.5, 0.625, 0.6875, 0.75, 0.8, 0.875, 1.0])),
               scale=8 if use_default else int(rng.choice([2, 4, 8, 16])), app=bool(rng.random() < 0.25),
               beta=0.0 if use_default else float(rng.choice([0.0, 0.125, 0.25, 0.3125, 0.375, 0.5, 0.77])))
    if i % 50 == 49:
        print("%d cases ok" % (i + 1), flush=True)
print("fuzz ok:", N)
