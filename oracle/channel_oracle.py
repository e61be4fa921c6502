"""CPU ORACLE (test infrastructure) for the fused channel kernel nrldpc_awgn_llr_dev.

numpy restatement, in float64, of what the kernel computes: NRModulator.m:73-81 (TS 38.211 5.1 maps, unit average
power), complex AWGN of variance N0 = 10^(-EsN0/10) (plot_BLER_vs_SNR.m:50,105-106) drawn from Philox-4x32-10
(counter = global symbol index div 2, key = seed; words 0,1 of the block feed the even symbol of the pair, words 2,3 the
odd one) through Box-Muller on 24-bit uniforms, and NRDemodulator.m:76-84's exact
log-likelihood ratios.  The kernel works in float32 with the device's logf / sincosf / expf, so tests compare within a
stated tolerance, not bit for bit.  Only tests/ may import this module.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter_lo, counter_hi, seed):
    """Philox-4x32-10 (Salmon et al., SC'11) for counters (lo, hi, 0, 0) and key (seed_lo, seed_hi): [n][4] uint32."""
    c0 = np.asarray(counter_lo, np.uint64) & M32
    c1 = np.asarray(counter_hi, np.uint64) & M32
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def noise(n_sym, seed, first_symbol, N0):
    """Complex noise of variance N0 for symbols first_symbol .. first_symbol + n_sym - 1."""
    s = np.uint64(first_symbol) + np.arange(n_sym, dtype=np.uint64)
    c = s >> np.uint64(1)
    r = philox4x32_10(c & M32, c >> np.uint64(32), int(seed))
    odd = (s & np.uint64(1)).astype(bool)
    w1 = np.where(odd, r[:, 2], r[:, 0])
    w2 = np.where(odd, r[:, 3], r[:, 1])
    u1 = ((w1 >> 8).astype(np.float64) + 0.5) / 16777216.0
    u2 = ((w2 >> 8).astype(np.float64) + 0.5) / 16777216.0
    rad = np.sqrt(-2.0 * np.log(u1)) * np.sqrt(N0 / 2.0)
    ang = 2.0 * np.pi * u2
    return rad * np.cos(ang) + 1j * rad * np.sin(ang)


def pam_levels(nb):
    """TS 38.211 5.1.3-5.1.5: rail amplitude from nb bits, sign bit first (16QAM: (1-2b0)(2-(1-2b2)), ...)."""
    amps = np.zeros(1 << nb)
    for code in range(1 << nb):
        bits = [(code >> (nb - 1 - i)) & 1 for i in range(nb)]
        x = 1.0
        for j, b in enumerate(reversed(bits[1:]), start=1):
            x = float(1 << j) - (1 - 2 * b) * x
        amps[code] = (1 - 2 * bits[0]) * x
    return amps


def awgn_llr(g, Q_m, EsN0_dB, seed, first_symbol=0):
    """bits g (flat, multiple of Q_m) -> exact LLRs (float64, positive = bit 0) of the noisy symbols."""
    g = np.asarray(g, np.uint8).reshape(-1, Q_m)
    N0 = 10.0 ** (-EsN0_dB / 10.0)
    w = noise(g.shape[0], seed, first_symbol, N0)
    if Q_m == 1:  # PSK order 2 at phase pi/4: LLR = 4 Re(rx e^{-j pi/4}) / N0
        rx = (1 - 2.0 * g[:, 0]) * np.exp(1j * np.pi / 4) + w
        return (4.0 * np.real(rx * np.exp(-1j * np.pi / 4)) / N0)[:, None].reshape(-1)
    nb = Q_m // 2
    amps = pam_levels(nb)
    pts = amps / np.sqrt(2.0 * np.mean(amps ** 2))
    codes = np.arange(1 << nb)
    cbits = (codes[:, None] >> np.arange(nb - 1, -1, -1)[None, :]) & 1
    wi = sum(g[:, 2 * k].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    wq = sum(g[:, 2 * k + 1].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    out = np.empty((g.shape[0], Q_m))
    for rail, y in ((0, pts[wi] + w.real), (1, pts[wq] + w.imag)):
        metric = -((y[:, None] - pts[None, :]) ** 2) / N0
        for k in range(nb):
            m0 = np.where(cbits[:, k] == 0, metric, -np.inf)
            m1 = np.where(cbits[:, k] == 1, metric, -np.inf)
            out[:, 2 * k + rail] = np.logaddexp.reduce(m0, axis=1) - np.logaddexp.reduce(m1, axis=1)
    return out.reshape(-1)
