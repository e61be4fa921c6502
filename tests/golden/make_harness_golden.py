#!/usr/bin/env python3
"""Generate tests/golden/harness_golden.json: the result files plot_BLER_vs_SNR writes (plot_BLER_vs_SNR.m:79,165:
one '%f\\t%e' line per finished Es/N0 point) and those plot_SNR_vs_A writes (plot_SNR_vs_A.m:80,186: one '%d\\t%f'
line per information block length) for small seeded runs of the host harness.

The harness draws payloads and noise from numpy's PCG64 (stable across numpy versions) and the decoder core is
bit-exact between the HIP kernels and the CPU oracle, so the whole Monte-Carlo run -- which SNR points exist, how many
blocks each took, every BLER digit -- is reproducible: generated here with oracle-backed encoder / decoder objects
(no GPU), reproduced on the GPU by tests/test_harness_gpu.py::test_result_files_match_committed_fixture.
MATLAB's own RNG streams cannot be reproduced, so this pins the build's harness, not MATLAB's
realisations.   Run from the repo root:  python tests/golden/make_harness_golden.py
"""
import importlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O  # noqa: E402
from make_step_golden import OracleCodec  # noqa: E402

pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")

RUNS = [  # keyword arguments of plot_BLER_vs_SNR (plot_BLER_vs_SNR.m:1,30-42)
    dict(A=100, R=1 / 3, BG=2, Modulation="QPSK", rv_id_sequence=[0], iterations=10, target_block_errors=25,
         target_BLER=2e-2, EsN0_start=-1.0, EsN0_delta=0.5, seed=11, batch=64),
    dict(A=[40, 300], R=0.5, BG=2, Modulation="16QAM", rv_id_sequence=[0, 2], iterations=8, target_block_errors=15,
         target_BLER=1e-1, EsN0_start=2.0, EsN0_delta=0.5, seed=3, batch=32),
]


RUNS_SNR_VS_A = [  # keyword arguments of plot_SNR_vs_A (plot_SNR_vs_A.m:1,38-48)
    dict(A=[40, 100, 300], R=[0.5, 1 / 3], BG=2, Modulation="QPSK", rv_id_sequence=[0], iterations=8, target_block_errors=12,
         target_BLER=1e-1, EsN0_start=-2.0, EsN0_delta=0.5, seed=5, batch=24),
    dict(A=[-1, 200], R=0.75, BG=2, Modulation="16QAM", rv_id_sequence=[0, 3], iterations=6, target_block_errors=10,
         target_BLER=2e-1, EsN0_start=4.0, EsN0_delta=1.0, seed=2, batch=16),  # A=-1 is refused by the objects: skipped (:165-172)
]


class _OracleEncCodec:
    def __init__(self, bg, Z):
        self.bg, self.Z = bg, Z

    def encode(self, info):
        return O.encode(self.bg, self.Z, info)

    def close(self):
        pass


def oracle_encoder(**kw):
    enc = pkg.NRLDPCEncoder(**kw)

    def setup():
        enc.validate()
        enc._codec = _OracleEncCodec(enc.BG, enc.Z_c)
        object.__setattr__(enc, "_locked", True)
    enc._setup = setup
    return enc


def oracle_decoder(**kw):
    dec = pkg.NRLDPCDecoder(**kw)

    def make(n_layers):
        dec._codec = OracleCodec(dec.BG, dec.Z_c, dec._setup_iterations, n_layers)
        dec._codec_layers = n_layers
    dec._make_codec = make
    return dec


def run_all(enc_factory=None, dec_factory=None):
    """{run index: {file name: text}} with the given System-object factories (None = the product's GPU-backed ones)."""
    saved = H.NRLDPCEncoder, H.NRLDPCDecoder
    if enc_factory:
        H.NRLDPCEncoder, H.NRLDPCDecoder = enc_factory, dec_factory
    try:
        out = {}
        for i, kw in enumerate(RUNS):
            with tempfile.TemporaryDirectory() as d:
                H.plot_BLER_vs_SNR(results_dir=d, **kw)
                out[str(i)] = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))}
        for i, kw in enumerate(RUNS_SNR_VS_A):
            with tempfile.TemporaryDirectory() as d:
                H.plot_SNR_vs_A(results_dir=d, **kw)
                out["snr_vs_a_%d" % i] = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))}
        return out
    finally:
        H.NRLDPCEncoder, H.NRLDPCDecoder = saved


if __name__ == "__main__":
    res = run_all(oracle_encoder, oracle_decoder)
    for i, files in res.items():
        for f, txt in files.items():
            print(i, f)
            print(txt, end="")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_golden.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
