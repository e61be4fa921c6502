#!/usr/bin/env python3
"""Generate tests/golden/bler_ref.npz: the sum-product side of every BLER comparison of tests/test_bler_gap_gpu.py.

oracle/orc_decode_bp_flood (flooding sum-product in double with the parity-check stop: the stand-in for the reference's
comm.LDPCDecoder, NRLDPCDecoder.m:120) on the seeded inputs of tests/bler_cases.py, every case and Es/N0 point: per block
whether its K' payload bits came out wrong, the mean number of sweeps, and a CRC of the LLRs.  CPU only; about twenty minutes on
eight cores.  tests/test_bler_ref.py re-computes a slice of every stored run on the CPU suite.
Run from the repo root:  python tests/golden/make_bler_ref.py [key-prefix ...]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import bler_cases as BC  # noqa: E402


def main():
    path = os.path.join(HERE, "bler_ref.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    only = sys.argv[1:]
    cache = {}
    nth = os.cpu_count() or 1
    for key, factory, case, bg, Z, nl, cap, snr in BC.runs():
        if only and not any(key.startswith(p) for p in only):
            continue
        if not only and key + "/err" in out:  # a resumed run: what is stored stays (name a prefix to recompute it)
            continue
        ck = (factory.__name__, case[0])
        if ck not in cache:
            cache.clear()
            cache[ck] = factory(case, O.encode)
        inp = cache[ck]
        llr = inp.llr_at(snr)
        t0 = time.time()
        hb, ib = O.decode_bp_flood(bg, Z, llr, cap, n_layers=nl, nthreads=nth)
        err = (hb[:, :inp.Kp] != inp.info[:, :inp.Kp]).any(1)
        out[key + "/err"] = np.packbits(err)
        out[key + "/sweeps"] = np.float64(ib.mean())
        out[key + "/llr_crc"] = np.uint32(BC.llr_crc(llr))
        print("%-60s BLER %.5f  mean sweeps %.2f  %.1f s" % (key, err.mean(), ib.mean(), time.time() - t0), flush=True)
        np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out) // 3, "runs")


if __name__ == "__main__":
    main()
