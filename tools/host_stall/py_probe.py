"""Where does the alternating 25-35 ms wait of the byte-per-bit host entry come from?  The C++ probe (stall_probe.cpp: output array
allocated once) does not show it; bench.py's e2e leg (a fresh numpy output array per call) does.  Variants, same library, same batch:
  fresh     output = np.empty(...) inside every call (Codec.decode: what bench.py measured)
  reuse     one output array for all calls (ctypes call straight into it)
  fresh+del fresh output, the previous one dropped BEFORE the call instead of after it
usage: py_probe.py <variant> <dtype f16|f64> <reps> [notorch]"""
import ctypes as C
import importlib
import os
import resource
import sys
import time

import numpy as np

variant, dts, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
if len(sys.argv) > 4 and sys.argv[4] == "notorch":
    sys.modules["torch"] = None  # _capi imports torch first when it can (one HIP runtime per process); not here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
dt = {"f16": np.float16, "f64": np.float64, "f32": np.float32}[dts]
B, Z = 4096, 384
c = pkg.Codec(1, Z, max_iter=25, early_term=False, llr_dtype=dt)
rng = np.random.default_rng(0)
llr = (4.0 + 2.8 * rng.standard_normal((B, c.N_cw), dtype=np.float32)).astype(dt)
out = np.empty((B, c.K), np.uint8)
out[:] = 1
h = None
lib = c._lib
for r in range(reps):
    f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    t0 = time.perf_counter()
    if variant == "reuse":
        pkg._capi.check(lib.nrldpc_decode(c._h, llr.ctypes.data_as(C.c_void_p), B, out.ctypes.data_as(C.c_void_p), None, None))
        h = out
    elif variant == "fresh":
        h = c.decode(llr)
    else:
        h = None
        h = c.decode(llr)
    t1 = time.perf_counter()
    f1 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    print("call %2d %s %s  %8.3f ms  minor faults %d  out@%x" % (r, variant, dts, (t1 - t0) * 1e3, f1 - f0, h.ctypes.data), flush=True)
assert int(h.sum()) == 0
c.close()
