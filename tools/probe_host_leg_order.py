#!/usr/bin/env python3
"""Round 6 probe: the host path's "6 ms mode" (a 4096-codeword call of singles taking 6.0-6.5 ms with 3.1-3.5 ms of it waiting for the
device, against 4.0 with 1.2-1.7) -- does it depend on how many codecs (= pairs of HIP streams) the process has created before?
argv: a comma list of dtypes; every entry creates a codec, runs 1 + 5 calls, reports, closes (or keeps it open with a trailing '+')."""
import importlib, json, os, sys, time
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
DT = {"f16": np.float16, "f32": np.float32, "f64": np.float64}
keep = []
for i, name in enumerate(sys.argv[1].split(",")):
    hold = name.endswith("+")
    dt = DT[name.rstrip("+")]
    x = x16.astype(dt)
    c = nrldpc.Codec(BG, Z, max_iter=25, n_layers=0, early_term=False, llr_dtype=dt)
    buf = np.zeros((B, (K + 7) // 8), np.uint8)
    c.decode_packed(x, out=buf)
    ts, ph = [], []
    for _ in range(5):
        t0 = time.perf_counter(); c.decode_packed(x, out=buf); ts.append((time.perf_counter() - t0) * 1e3); ph.append(c.last_host_phases())
    j = int(np.argsort(ts)[2])
    print(json.dumps({"order": i, "dtype": name, "ms": [round(t, 2) for t in ts], "q": round(ph[j]["copy_quantise_ms"], 2), "wait": round(ph[j]["wait_device_ms"], 2)}), flush=True)
    if hold:
        keep.append(c)
    else:
        c.close()
