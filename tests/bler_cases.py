"""Inputs of the BLER comparisons (tests/test_bler_gap_gpu.py), shared with the generator of the committed sum-product results
(tests/golden/make_bler_ref.py) and with the CPU test that pins them (tests/test_bler_ref.py).

The sum-product side of every comparison -- oracle/orc_decode_bp_flood, the stand-in for the reference's comm.LDPCDecoder
(NRLDPCDecoder.m:120) -- is a pure function of (case, Es/N0): same seeds, same payloads, same noise.  It is what the GPU suite's
wall time was made of (4096 blocks x 50 sweeps of double-precision sum-product on the GPU box's 16 granted cores: 727 s of a
1200 s limit in round 4), so its per-block outcomes are computed ONCE by the committed generator and stored in
tests/golden/bler_ref.npz; the GPU test decodes the same LLRs on the GPU and compares against the stored outcomes, after checking
that the LLRs it built are the ones the generator saw (a CRC of the first blocks' bytes)."""
import zlib

import numpy as np

BG_DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}

# name, bg, Z, K' (payload + CRC bits), E (transmitted bits, rv0), active layers, iteration cap, Es/N0 grid, blocks
CASES = [
    ("cfg1 BG2 A=100 R=1/3 QPSK 10it", 2, 20, 116, 300, 12, 10, [0.0, 0.5, 1.0, 1.5, 2.0], 4096),
    ("cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.6, -1.5, -1.4, -1.3, -1.2], 1024),
    ("cfg3 BG2 Z=384 R=1/5 25it", 2, 384, 3840, 19120, 42, 25, [-4.2, -4.1, -4.0, -3.9, -3.8], 512),
    ("cfg3 BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.5, -1.4, -1.3, -1.2, -1.1], 512),
    ("cfg3 BG2 Z=384 R=2/3 25it", 2, 384, 3840, 5736, 7, 25, [2.6, 2.8, 3.0, 3.2, 3.4], 512),
    ("cfg5 BG1 Z=384 R=8/9 25it", 1, 384, 8448, 9478, 5, 25, [5.8, 6.0, 6.2, 6.4, 6.6], 512),
]
# name (as in CASES), bg, Z, K', E, layers, iteration cap, grid at equal caps, grid of the 50-sweep sum-product reference, blocks
CASES_1E2 = [
    ("cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.35, -1.30, -1.25], [-1.65, -1.60, -1.55], 4096),
    ("cfg3 BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.30, -1.20, -1.10], [-1.60, -1.50, -1.40], 4096),
    ("cfg5 BG1 Z=384 R=8/9 25it", 1, 384, 8448, 9478, 5, 25, [6.1, 6.2, 6.3, 6.4], [6.0, 6.1, 6.2, 6.3], 4096),
]
# BLER 1e-3 -- where the reference's sweep stops (target_BLER, plot_BLER_vs_SNR.m:38) -- at equal caps, on 16384 blocks:
# name, bg, Z, K', E, layers, iteration cap, grid of the GPU decoder, grid of the sum-product oracle, blocks
CASES_1E3 = [
    ("cfg2 headline BG1 Z=384 R=1/3 25it", 1, 384, 8448, 25272, 46, 25, [-1.30, -1.25, -1.20, -1.15], [-1.30, -1.25, -1.20], 16384),
    ("cfg3 BG2 Z=384 R=1/3 25it", 2, 384, 3840, 11472, 22, 25, [-1.25, -1.20, -1.15, -1.10], [-1.15, -1.10, -1.05, -1.00], 16384),
    ("cfg1 BG2 A=100 R=1/3 QPSK 10it", 2, 20, 116, 300, 12, 10, [1.4, 1.6, 1.8, 2.0], [2.0, 2.2, 2.4, 2.6], 65536),
]
# name, bg, Z, K', E, layers, grid of the GPU decoder, grid of the sum-product oracle, blocks
CASES_50 = [
    ("cfg2 headline BG1 Z=384 R=1/3 50it", 1, 384, 8448, 25272, 46, [-1.50, -1.45, -1.40, -1.35, -1.30], [-1.70, -1.65, -1.60, -1.55], 4096),
    ("cfg3 BG2 Z=384 R=1/3 50it", 2, 384, 3840, 11472, 22, [-1.55, -1.45, -1.35, -1.25, -1.15], [-1.75, -1.65, -1.55, -1.45, -1.35], 4096),
]


class Inputs:
    """Payloads, codewords and the noise of one case; llr_at(snr) = the cw_tilde of NRLDPCDecoder.m:262-264 at that Es/N0."""

    def __init__(self, seed_text, bg, Z, Kp, E, nblk, encode, noise_f32, first=None):
        """first: only the first `first` blocks (the generator fills its arrays block by block, so they are the same blocks)."""
        rows, cols, kb = BG_DIMS[bg]
        self.bg, self.Z, self.Kp, self.E, self.K = bg, Z, Kp, E, kb * Z
        rng = np.random.default_rng(zlib.crc32(seed_text.encode()))
        self.info = rng.integers(0, 2, (nblk, self.K), dtype=np.uint8)
        self.info[:, Kp:] = 0
        if first is not None:
            self.info = self.info[:first]
        self.cw = encode(bg, Z, self.info)
        self.noise = rng.standard_normal(self.cw.shape)
        if noise_f32:
            self.noise = self.noise.astype(np.float32)

    def llr_at(self, snr):
        Z, K, Kp, E = self.Z, self.K, self.Kp, self.E
        mu = 2 * 10 ** (snr / 10)                    # QPSK, N0 = 10^(-EsN0/10)  (plot_BLER_vs_SNR.m:105-106)
        llr = ((1 - 2.0 * self.cw) * mu + np.sqrt(2 * mu) * self.noise).astype(np.float64)
        llr[:, : 2 * Z] = 0                          # punctured systematic columns (NRLDPCDecoder.m:262)
        llr[:, 2 * Z + E + (K - Kp):] = 0            # beyond the E transmitted non-filler bits (k0 = 0)
        llr[:, Kp:K] = np.inf                        # fillers (NRLDPCDecoder.m:264)
        return llr


def inputs_gap(case, encode, first=None):
    name, bg, Z, Kp, E, nl, iters, snrs, nblk = case
    return Inputs(name, bg, Z, Kp, E, nblk, encode, False, first)


def inputs_1e2(case, encode, first=None):
    name, bg, Z, Kp, E, nl, iters, snrs, snrs50, nblk = case
    return Inputs(name + " 1e-2", bg, Z, Kp, E, nblk, encode, True, first)


def inputs_1e3(case, encode, first=None):
    name, bg, Z, Kp, E, nl, iters, snrs, snrs_bp, nblk = case
    return Inputs(name + " 1e-3", bg, Z, Kp, E, nblk, encode, True, first)


def inputs_50(case, encode, first=None):
    name, bg, Z, Kp, E, nl, snrs, snrs_bp, nblk = case
    return Inputs(name, bg, Z, Kp, E, nblk, encode, True, first)


def runs():
    """Every sum-product run of the suite: (key, inputs factory, case, bg, Z, layers, cap, snr)."""
    out = []
    for c in CASES:
        for snr in c[7]:
            out.append(("gap/%s/%g" % (c[0], snr), inputs_gap, c, c[1], c[2], c[5], c[6], snr))
    for c in CASES_1E2:
        for snr in c[7]:
            out.append(("1e2/%s/%g" % (c[0], snr), inputs_1e2, c, c[1], c[2], c[5], c[6], snr))
        for snr in c[8]:
            out.append(("1e2_50/%s/%g" % (c[0], snr), inputs_1e2, c, c[1], c[2], c[5], 50, snr))
    for c in CASES_50:
        for snr in c[7]:
            out.append(("50/%s/%g" % (c[0], snr), inputs_50, c, c[1], c[2], c[5], 50, snr))
    for c in CASES_1E3:
        for snr in c[8]:
            out.append(("1e3/%s/%g" % (c[0], snr), inputs_1e3, c, c[1], c[2], c[5], c[6], snr))
    return out


def llr_crc(llr):
    """Ties a stored result to the LLRs it was computed from: CRC-32 of the first 8 blocks' bytes."""
    return zlib.crc32(np.ascontiguousarray(llr[:8]).tobytes())


class Ref:
    """tests/golden/bler_ref.npz: per run the block-error indicator of every block (bit-packed), the mean number of sweeps and
    the CRC of the LLRs."""

    def __init__(self, path):
        self.d = np.load(path) if path else None

    def get(self, key, llr, Kp, info, nblk):
        """(block errors [nblk] bool, mean sweeps) of the stored sum-product run, or None when it is not stored / was computed
        from other LLRs."""
        if self.d is None or key + "/err" not in self.d.files:
            return None
        if int(self.d[key + "/llr_crc"]) != llr_crc(llr):
            return None
        err = np.unpackbits(self.d[key + "/err"])[:nblk].astype(bool)
        return err, float(self.d[key + "/sweeps"])
