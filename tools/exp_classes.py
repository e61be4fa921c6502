#!/usr/bin/env python3
"""Experiment (round 6): layer-count CLASSES -- the run-time-prefix kernels over the tables of the first c rows only (nl_class, nrldpc_device.h).
Parity of a class library (NRLDPC_LIB, built with -DNRLDPC_Z64_RTC=c by tools/exp_units.sh) against the oracle for every layer count <= c:
    NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_LIB=exp_libs/lib_rtc24.so python tools/exp_classes.py 24 1,384 2,384"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
T = importlib.import_module("test_decode_gpu")
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
cls = int(sys.argv[1])
rng = np.random.default_rng(17)
for arg in sys.argv[2:]:
    bg, Z = (int(x) for x in arg.split(","))
    for nl in sorted({4, 5, 7, cls - 1, cls} | set(range(6, cls, 5))):
        if nl < 4 or nl > cls:
            continue
        w = T._waterfall_esn0(bg, nl)
        try:
            T.run_case(pkg, orc, rng, bg, Z, 3, w - 2.5, 2, nl=nl, et=False, app=False)
            T.run_case(pkg, orc, rng, bg, Z, 3, w, 25, nl=nl, et=False, app=False)
            T.run_case(pkg, orc, rng, bg, Z, 5, w + 0.2, 14, nl=nl, et=True, app=False)
            print("class %d BG%d Z=%d nl=%d parity ok" % (cls, bg, Z, nl), flush=True)
        except AssertionError as e:
            print("class %d BG%d Z=%d nl=%d PARITY FAIL %s" % (cls, bg, Z, nl, str(e)[:200]), flush=True)
