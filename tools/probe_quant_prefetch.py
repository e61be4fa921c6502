#!/usr/bin/env python3
"""Round 6 experiment: does a software prefetch in the AVX-512 host quantiser lift the copy threads' streaming rate?  The host's DRAM share
drifts 2x over seconds (profiles/r06_f64_leg_probe.txt), so the paths alternate CALL BY CALL in one process on one array:
NRLDPC_HOST_QUANT_PATH = 2 (shipped), 3 (prefetch 1 KB ahead), 4 (4 KB ahead); the copy / quantise phase of every call
(nrldpc_last_host_phases) is what is compared."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
nrldpc = importlib.import_module("ldpc-3gpp-matlab_amd")
BG, Z, B = 1, 384, 4096
K, NCW = 22 * Z, 68 * Z
rng = np.random.default_rng(1)
x16 = (rng.standard_normal((B, NCW)) * 2 + 1).astype(np.float16)
x16[:, : 2 * Z] = 0
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:  # noqa: BLE001
        return (0, 0)


PATHS = tuple(int(x) for x in os.environ.get("PROBE_PATHS", "2,3,4").split(","))
for dt in (np.float64, np.float16, np.float32):
    x = x16.astype(dt)
    c = nrldpc.Codec(BG, Z, max_iter=25, n_layers=0, early_term=False, llr_dtype=dt)
    buf = np.zeros((B, (K + 7) // 8), np.uint8)
    c.decode_packed(x, out=buf)
    res = {p_: [] for p_ in PATHS}
    wall = []
    th0 = throttled()
    import time
    for rep in range(12):
        for path in PATHS:
            os.environ["NRLDPC_HOST_QUANT_PATH"] = str(path)
            t0 = time.perf_counter()
            c.decode_packed(x, out=buf)
            wall.append((time.perf_counter() - t0) * 1e3)
            res[path].append(c.last_host_phases()["copy_quantise_ms"])
    th1 = throttled()
    os.environ.pop("NRLDPC_HOST_QUANT_PATH", None)
    c.close()
    print(json.dumps({"dtype": np.dtype(dt).name, "spin_us": os.environ.get("NRLDPC_HOST_SPIN_US", "default"), "wall_ms_median": float(np.median(wall)), "wall_ms_max": float(np.max(wall)),
                      "throttled_periods_during": th1[0] - th0[0], "throttled_ms_during": (th1[1] - th0[1]) / 1e3, "copy_quantise_ms_median": {str(p): float(np.median(v)) for p, v in res.items()},
                      "min": {str(p): float(np.min(v)) for p, v in res.items()}, "all": {str(p): [round(t, 2) for t in v] for p, v in res.items()}}), flush=True)
