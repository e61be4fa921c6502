#!/usr/bin/env python3
"""Generate tests/golden/step_golden.npz: sequences of NRLDPCDecoder.step() calls with decoder state between them
(incremental redundancy, sticky code-block CRC flags, CBGTI) and the outputs the reference's state machine gives
(NRLDPCDecoder.m:133-140, 236-239, 283-316, 336-339, 343-356).

Expected values come from the host mirror of the System object (ldpc-3gpp-matlab_amd/decoder.py, a line-by-line
restatement of those lines) with the decoder CORE replaced by the CPU oracle (orc_decode_onmsq, bit-identical to the
HIP kernels), so the file can be made -- and re-checked by the CPU test suite -- without a GPU.  The reference holds no
decoder vectors; these pin the build's step() semantics.

Scenarios (BG2, A = 3842 -> C = 2 code blocks with CB-CRC24B, QPSK, G = 5000 per transmission, 3 transport blocks):
  harq_jam    I_HARQ = 1, rv 0,2,3,1.  Transport block 0: code block 0 arrives clean and code block 1 as noise in the
              first transmission; in the second, code block 0 is jammed (large wrong-sign LLRs that wreck its soft
              buffer) while code block 1 arrives clean.  The reference returns a_hat after step 2: block 0 was kept in
              b_hat_buffer with its sticky pass flag (:286-287,305).  A stateless CRC stage returns [] there.
  harq_cbgti  I_HARQ = 1; step 2 retransmits only code block 1 (CBGTI = [0]): block 0 is neither decoded into b_hat
              nor allowed to clear its flag (:304).
  noharq      I_HARQ = 0, rv 0 in every step, same clean / noise / jammed pattern: b_hat starts from zeros every step
              (:289) while the pass flags stay sticky until reset() (:280,315) -- transport block 0 ends with both flags
              set but block 0's segment zero, so its TB CRC fails: the reference's behaviour, kept as is.
Run from the repo root:  python tests/golden/make_step_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
KW = dict(BG=2, A=3842, Q_m=2)
G_TX = 5000
ITERS = 12


class OracleCodec:
    """Stands in for _capi.Codec inside the host mirror: same check-node rule, decoded by the CPU oracle."""

    def __init__(self, bg, Z, max_iter, n_layers):
        self.bg, self.Z, self.max_iter, self.n_layers = bg, Z, max_iter, n_layers
        self.alpha, self.beta = pkg.default_rule(bg, n_layers)

    def decode(self, llr, want_iters=False):
        h, it = O.decode_nmsq(self.bg, self.Z, np.asarray(llr, np.float32).astype(np.float64), self.max_iter,
                              n_layers=self.n_layers, early_term=True, alpha=self.alpha, beta=self.beta * 8)
        return (h, it) if want_iters else h

    def last_layers(self):
        return self.n_layers

    def close(self):
        pass


def oracle_decoder(**kw):
    """NRLDPCDecoder host mirror whose stage 4 runs on the oracle instead of the GPU."""
    dec = pkg.NRLDPCDecoder(**kw)

    def make(n_layers):
        dec._codec = OracleCodec(dec.BG, dec.Z_c, dec._setup_iterations, n_layers)
        dec._codec_layers = n_layers
    dec._make_codec = make
    return dec


def transmissions(rng, scenario):
    """List of steps: (rv_id, CBGTI, G, g_tilde [3][G] float16).  Encoding on the CPU (oracle encoder)."""
    enc = pkg.NRLDPCEncoder(G=G_TX, **KW)
    enc.validate()
    a = rng.integers(0, 2, (3, enc.A), dtype=np.uint8)
    c = enc.code_block_segmentation(enc.crc_calculation(a))
    cw = np.stack([O.encode(enc.BG, enc.Z_c, c[t]) for t in range(3)])[:, :, 2 * enc.Z_c:]
    steps = []
    plan = [(0, []), (2, [0] if scenario == "harq_cbgti" else []), (3, []), (1, [])]
    if scenario == "noharq":  # every transmission has to stand alone: rv 0 each time
        plan = [(0, []), (0, []), (0, []), (0, [])]
    for n, (rv, cbgti) in enumerate(plan):
        enc.rv_id, enc.CBGTI = rv, cbgti
        enc.G = G_TX if not cbgti else G_TX // 2
        g = enc.rate_match(cw).astype(np.float64)                       # [3][G]
        E = enc.E_r
        mu = 2 * 10 ** (3.0 / 10)
        llr = (1 - 2 * g) * mu + np.sqrt(2 * mu) * rng.standard_normal(g.shape)
        seg0 = slice(0, E[0])
        seg1 = slice(E[0], E[0] + E[1])
        if n == 0:
            llr[0, seg0] = (1 - 2 * g[0, seg0]) * 12.0                  # TB 0: block 0 clean ...
            llr[0, seg1] = 0.3 * rng.standard_normal(E[1])              # ... block 1 noise only
        if n == 1 and not cbgti:
            llr[0, seg0] = -(1 - 2 * g[0, seg0]) * 60.0                 # TB 0: block 0 jammed (wrong sign, strong)
            llr[0, seg1] = (1 - 2 * g[0, seg1]) * 12.0                  # block 1 clean
        if n == 1 and cbgti:
            llr[0] = (1 - 2 * g[0]) * 12.0                              # only block 1 is on the air, clean
        llr[2] = 0.2 * rng.standard_normal(g.shape[1])                  # TB 2: hopeless in every transmission
        steps.append((rv, cbgti, int(enc.G), llr.astype(np.float16)))
    return a, steps


def run(scenario, a, steps, make_decoder):
    """Feeds the steps to a decoder built by make_decoder(**kw); returns per-step (a_hat [3][A], ok [3], cb_pass [3][C])."""
    dec = make_decoder(I_HARQ=0 if scenario == "noharq" else 1, iterations=ITERS, G=G_TX, **KW)
    out = []
    for rv, cbgti, G, llr in steps:
        dec.rv_id, dec.CBGTI, dec.G = rv, cbgti, G
        a_hat, ok = dec.step_batch(llr.astype(np.float64))
        out.append((a_hat.copy(), ok.copy(), dec.code_block_CRC_passed.copy()))
    dec.release()
    return out


def main():
    out = {}
    for k, scenario in enumerate(("harq_jam", "harq_cbgti", "noharq")):
        rng = np.random.default_rng(77 + k)
        a, steps = transmissions(rng, scenario)
        res = run(scenario, a, steps, oracle_decoder)
        out[scenario + "/a"] = a
        out[scenario + "/plan"] = np.frombuffer(json.dumps([(rv, cb, G) for rv, cb, G, _ in steps]).encode(), dtype=np.uint8)
        for n, ((rv, cb, G, llr), (a_hat, ok, passed)) in enumerate(zip(steps, res)):
            out["%s/g_tilde_%d" % (scenario, n)] = llr
            out["%s/a_hat_packed_%d" % (scenario, n)] = np.packbits(a_hat, axis=1)
            out["%s/ok_%d" % (scenario, n)] = ok.astype(np.uint8)
            out["%s/cb_pass_%d" % (scenario, n)] = passed.astype(np.uint8)
            print(scenario, "step", n, "rv", rv, "CBGTI", cb, "ok", ok.astype(int), "cb_pass", passed.tolist(),
                  "a_hat == a", [(a_hat[t] == a[t]).all() for t in range(3)])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
