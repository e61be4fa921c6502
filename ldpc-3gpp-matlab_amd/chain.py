"""TS 38.212 stages either side of the LDPC core, as index maps + CRC (host logic, numpy).

These are the reference's interpreted per-element loops (NRLDPCEncoder.m:70-256,
NRLDPCDecoder.m:143-242, 271-340) restated as whole-array gathers.  They sit OUTSIDE the hot path
("step before / steps after"); the hot path itself is the HIP kernel
behind _capi.Codec.  Device-side versions of rate recovery and the CRC check are rows N1/N2.
"""
import numpy as np

from .nrldpc import NRLDPC


# ---------------------------------------------------------------------------------------------
# CRC (comm.CRCGenerator / comm.CRCDetector with zero initial state, no reflection, no final XOR;
# polynomials from get_3gpp_crc_polynomial.m:3-14)
# ---------------------------------------------------------------------------------------------
_TABLES = {}


def _crc_table(poly, L):
    key = (poly, L)
    if key not in _TABLES:
        top, mask = 1 << (L - 1), (1 << L) - 1
        tab = np.zeros(256, np.uint32)
        for b in range(256):
            r = b << (L - 8)
            for _ in range(8):
                r = (((r << 1) ^ poly) if (r & top) else (r << 1)) & mask
            tab[b] = r
        _TABLES[key] = tab
    return _TABLES[key]


def crc_bits(bits, poly, L):
    """CRC remainder (L bits, MSB first) of a 0/1 array; a 2-D input [batch][n] gives [batch][L]."""
    bits = np.asarray(bits, np.uint8)
    single = bits.ndim == 1
    bits = bits.reshape(1, -1) if single else bits
    pad = (-bits.shape[1]) % 8
    if pad:  # leading zeros do not change a zero-initialised CRC
        bits = np.concatenate([np.zeros((bits.shape[0], pad), np.uint8), bits], axis=1)
    data = np.packbits(bits, axis=1).astype(np.uint32)
    tab = _crc_table(poly, L)
    mask = np.uint32((1 << L) - 1)
    r = np.zeros(bits.shape[0], np.uint32)
    for k in range(data.shape[1]):  # byte-serial over the message, vectorised over the batch
        r = ((r << np.uint32(8)) & mask) ^ tab[((r >> np.uint32(L - 8)) ^ data[:, k]) & np.uint32(0xFF)]
    out = ((r[:, None] >> np.arange(L - 1, -1, -1, dtype=np.uint32)[None, :]) & np.uint32(1)).astype(np.uint8)
    return out[0] if single else out


# ---------------------------------------------------------------------------------------------
# Index maps
# ---------------------------------------------------------------------------------------------
def selection_index(p: NRLDPC, r: int):
    """idx[k] = position in d (0..N-1) of the k-th rate-matched bit of code block r
    (circular-buffer walk from k_0 skipping fillers: NRLDPCEncoder.m:186-195, NRLDPCDecoder.m:226-234)."""
    E = p.E_r[r]
    Z, N_cb, k0 = p.Z_c, p.N_cb, p.k_0
    pos = (k0 + np.arange(N_cb)) % N_cb
    lo_f, hi_f = max(int(p.K_prime) - 2 * Z, 0), p.K - 2 * Z  # fillers: NRLDPCDecoder.m:224
    pos = pos[(pos < lo_f) | (pos >= hi_f)]
    if E == 0:
        return np.zeros(0, np.int64)
    return pos[np.arange(E) % pos.size]


def interleave_index(E: int, Q_m: int):
    """perm with f[perm_f] = e[k]:  f(i + j*Q_m) = e(i*E/Q_m + j)  (NRLDPCEncoder.m:219-223).
    Returns fpos[k] = position in f of e[k]."""
    k = np.arange(E)
    rows = E // Q_m
    return (k % rows) * Q_m + k // rows


def g_to_d_maps(p: NRLDPC):
    """Per code block r: (offset of f_r in g, dpos[k] for k-th e bit, fpos[k])."""
    maps, off = [], 0
    for r, E in enumerate(p.E_r):
        maps.append((off, selection_index(p, r), interleave_index(E, p.Q_m) if E else np.zeros(0, np.int64)))
        off += E
    return maps
