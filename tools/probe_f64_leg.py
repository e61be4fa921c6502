#!/usr/bin/env python3
"""why does the MATLAB-doubles leg of bench.py's e2e (nrldpc_decode_packed on 856 MB of float64) lose to the
byte-per-bit leg on the same doubles in the same run, and to its own read bound?  Runs the two entry points on doubles in both
orders, with the input array (a) made fresh by astype() for every leg, as bench.py does, and (b) made once and shared, and prints
per-call wall times and the library's own phase times (nrldpc_last_host_phases) plus where the array's pages live (numa_maps)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (shares the HIP runtime, as in bench.py)
nrldpc = importlib.import_module("ldpc-3gpp-matlab_amd")
BG, Z, B, ITERS = 1, 384, 4096, 25
K, NCW = 22 * Z, 68 * Z


def pages_by_node(a):
    """NUMA node -> pages of the mapping that holds array a (from /proc/self/numa_maps)."""
    addr = a.__array_interface__["data"][0]
    try:
        best = None
        for line in open("/proc/self/numa_maps"):
            f = line.split()
            start = int(f[0], 16)
            if start <= addr:
                if best is None or start > best[0]:
                    best = (start, f)
        nodes = {t.split("=")[0]: int(t.split("=")[1]) for t in best[1] if t.startswith("N") and "=" in t}
        extra = [t for t in best[1] if t.startswith(("kernelpagesize", "huge", "anon"))]
        return {"nodes": nodes, "flags": extra}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def leg(name, x, packed, reps=7):
    c = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=x.dtype)
    buf = np.zeros((B, (K + 7) // 8 if packed else K), np.uint8)
    call = c.decode_packed if packed else c.decode
    call(x, out=buf)
    ts, ph = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        call(x, out=buf)
        ts.append((time.perf_counter() - t0) * 1e3)
        ph.append(c.last_host_phases())
    c.close()
    i = int(np.argsort(ts)[len(ts) // 2])
    print(json.dumps({"leg": name, "ms": [round(t, 2) for t in ts], "median_call_phases": ph[i], "input_pages": pages_by_node(x),
                      "caller_cpu": os.sched_getcpu() if hasattr(os, "sched_getcpu") else None}), flush=True)


def main():
    rng = np.random.default_rng(1)
    x16 = (rng.standard_normal((B, NCW)) * 2 + 1).astype(np.float16)
    x16[:, : 2 * Z] = 0
    print(json.dumps({"cpus": sorted(os.sched_getaffinity(0))[:4], "n_cpus": len(os.sched_getaffinity(0))}))
    for order in (("packed", "bytes"), ("bytes", "packed")):
        for kind in order:  # bench.py's way: a fresh astype() array per leg
            leg("fresh_array/%s first=%s" % (kind, order[0]), x16.astype(np.float64), kind == "packed")
    x = x16.astype(np.float64)
    for kind in ("packed", "bytes", "packed", "bytes"):
        leg("shared_array/%s" % kind, x, kind == "packed")
    # the same array after every page has been touched by THIS thread only once more (no migration expected): control
    x2 = np.empty_like(x); x2[:] = x
    for kind in ("packed", "bytes"):
        leg("copied_array/%s" % kind, x2, kind == "packed")
    # fp16 and single for scale
    leg("f16/packed", x16, True)
    leg("f32/packed", x16.astype(np.float32), True)


if __name__ == "__main__":
    main()
