#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-pointer entry point nrldpc_decode (what a MEX gateway calls):
pageable host arrays in, hard bits out, synchronous.  BG1 Z=384, early termination on (reference semantics),
QPSK/AWGN at Es/N0 = -0.5 dB, the library's default check-node rule.  min / median / max over RUNS calls per
point (the first call of a handle, which allocates the pinned slots, is reported separately).
    python tools/bench_host_path.py            -> gpurun_out/bench_host_path.json
NRLDPC_HOST_THREADS / NRLDPC_HOST_CHUNK_MB are read by the library once per process: a sweep over them re-executes
this script (python tools/bench_host_path.py --sweep)."""
import importlib, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RUNS = 12


def measure(batches=(1, 16, 256, 4096), dtypes=(np.float16, np.float32, np.float64)):
    pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
    bg, Z, cols, kb = 1, 384, 68, 22
    rng = np.random.default_rng(0)
    out = []
    enc = pkg.Codec(bg, Z, max_iter=1, llr_dtype=np.float32)
    for dt in dtypes:
        for B in batches:
            c = pkg.Codec(bg, Z, max_iter=25, early_term=True, llr_dtype=dt)
            info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
            cw = enc.encode(info)
            mu = 2.0 * 10 ** (-0.5 / 10)
            llr = ((1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * rng.standard_normal(cw.shape)).astype(np.float32)
            llr[:, : 2 * Z] = 0
            llr[:, 2 * Z + 25344:] = 0
            llr = np.ascontiguousarray(llr.astype(dt))
            t0 = time.perf_counter(); c.decode(llr); first = time.perf_counter() - t0
            ts = []
            for _ in range(RUNS):
                t0 = time.perf_counter()
                hard = c.decode(llr)
                ts.append(time.perf_counter() - t0)
            c.close()
            ts.sort()
            med = ts[len(ts) // 2]
            rec = {"llr_dtype": np.dtype(dt).name, "batch": B, "runs": RUNS, "ms_min": ts[0] * 1e3, "ms_median": med * 1e3,
                   "ms_max": ts[-1] * 1e3, "ms_first_call": first * 1e3, "info_Gbit_s_median": B * kb * Z / med / 1e9,
                   "host_threads": os.environ.get("NRLDPC_HOST_THREADS", "default"),
                   "chunk_mb": os.environ.get("NRLDPC_HOST_CHUNK_MB", "default"),
                   "block_errors": int((hard != info).any(1).sum())}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if "--sweep" in sys.argv:
        for th in ("4", "8", "16", "32"):
            for mb in ("16", "32", "64"):
                env = dict(os.environ, NRLDPC_HOST_THREADS=th, NRLDPC_HOST_CHUNK_MB=mb)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--big-only"], env=env)
    elif "--big-only" in sys.argv:
        measure(batches=(4096,), dtypes=(np.float16, np.float64))
    else:
        json.dump(measure(), open(os.path.join(ROOT, "gpurun_out", "bench_host_path.json"), "w"), indent=1)
