"""Monte-Carlo BLER harness: the control flow of the reference's plot_BLER_vs_SNR.m:104-171 with
transport blocks simulated in batches so that the decoder sees full GPU launches.

Per (BG, R, A) point it sweeps Es/N0 upward from EsN0_start in EsN0_delta steps until BLER <=
target_BLER, collecting target_block_errors errors per SNR (the "found_start" restart rule of :139-144
included), runs the HARQ loop over rv_id_sequence with I_HARQ = 1 and a reset per block (:120-137), and
writes `results/BLER_vs_SNR_<A>_<R>_<BG>_<Mod>_<iters>_<errs>_<EsN0start>_<seed>.txt` with one
`%f\\t%e` line per finished SNR (:79, :165).  MATLAB's RNG streams cannot be reproduced, so curves
compare statistically, never per realisation.

Modulation / demodulation restate NRModulator.m:73-81 / NRDemodulator.m:76-84: TS 38.211 Gray maps
(the reference's custom symbol tables), unit average power, exact log-likelihood ratios with the noise
variance N0 = 10^(-EsN0/10) (plot_BLER_vs_SNR.m:105-106).
"""
import os

import numpy as np

from ._capi import UnsupportedParameters
from .decoder import NRLDPCDecoder
from .encoder import NRLDPCEncoder

Q_M = {"BPSK": 1, "QPSK": 2, "16QAM": 4, "64QAM": 6, "256QAM": 8}


def _pam_levels(nb):
    """TS 38.211 5.1: amplitude of one I/Q rail from its nb bits (first bit = sign, then nested Gray):
    16QAM (1-2b0)(2-(1-2b2)); 64QAM (1-2b0)(4-(1-2b2)(2-(1-2b4))); 256QAM one level more."""
    pts = {}
    for code in range(1 << nb):
        bits = [(code >> (nb - 1 - i)) & 1 for i in range(nb)]
        x = 1.0
        for j, b in enumerate(reversed(bits[1:]), start=1):  # innermost bit first
            x = float(1 << j) - (1 - 2 * b) * x
        pts[code] = (1 - 2 * bits[0]) * x
    return pts


def _rail(nb):
    lv = _pam_levels(nb)
    codes = np.arange(1 << nb)
    amps = np.array([lv[c] for c in codes], np.float64)
    bits = ((codes[:, None] >> np.arange(nb - 1, -1, -1)[None, :]) & 1).astype(np.uint8)
    return amps, bits


def modulate(g, Q_m):
    """bits [..., G] -> complex symbols [..., G/Q_m], unit average power."""
    g = np.asarray(g, np.uint8)
    if Q_m == 1:  # comm.PSKModulator order 2, phase offset pi/4 (NRModulator.m:73)
        return (1 - 2.0 * g) * np.exp(1j * np.pi / 4)
    b = g.reshape(g.shape[:-1] + (-1, Q_m))
    nb = Q_m // 2
    amps, _ = _rail(nb)
    wi = sum(b[..., 2 * k].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    wq = sum(b[..., 2 * k + 1].astype(np.int64) << (nb - 1 - k) for k in range(nb))
    norm = np.sqrt(2.0 * np.mean(amps ** 2))
    return (amps[wi] + 1j * amps[wq]) / norm


def demodulate_llr(rx, Q_m, N0):
    """Exact LLRs (positive = bit 0) for the maps of `modulate`, complex noise variance N0."""
    if Q_m == 1:
        y = np.real(rx * np.exp(-1j * np.pi / 4))
        return 4.0 * y / N0
    nb = Q_m // 2
    amps, bits = _rail(nb)
    norm = np.sqrt(2.0 * np.mean(amps ** 2))
    pts = amps / norm
    out = np.empty(rx.shape + (Q_m,), np.float64)
    for rail, y in ((0, np.real(rx)), (1, np.imag(rx))):
        metric = -((y[..., None] - pts) ** 2) / N0            # [..., 2^nb]
        for k in range(nb):
            m0 = np.where(bits[:, k] == 0, metric, -np.inf)
            m1 = np.where(bits[:, k] == 1, metric, -np.inf)
            out[..., 2 * k + rail] = np.logaddexp.reduce(m0, axis=-1) - np.logaddexp.reduce(m1, axis=-1)
    return out.reshape(rx.shape[:-1] + (-1,))


def modulate_t(g, Q_m):
    """torch version of `modulate` (device tensors)."""
    import torch
    if Q_m == 1:
        return (1 - 2.0 * g.double()) * complex(np.cos(np.pi / 4), np.sin(np.pi / 4))
    nb = Q_m // 2
    amps, _ = _rail(nb)
    norm = float(np.sqrt(2.0 * np.mean(amps ** 2)))
    lut = torch.tensor(amps / norm, dtype=torch.float64, device=g.device)
    b = g.reshape(g.shape[:-1] + (-1, Q_m)).long()
    wi = sum(b[..., 2 * k] << (nb - 1 - k) for k in range(nb))
    wq = sum(b[..., 2 * k + 1] << (nb - 1 - k) for k in range(nb))
    return torch.complex(lut[wi], lut[wq])


def demodulate_llr_t(rx, Q_m, N0):
    """torch version of `demodulate_llr` (device tensors), exact log-MAP LLRs."""
    import torch
    if Q_m == 1:
        rot = complex(np.cos(np.pi / 4), -np.sin(np.pi / 4))
        return 4.0 * (rx * rot).real / N0
    nb = Q_m // 2
    amps, bits = _rail(nb)
    norm = float(np.sqrt(2.0 * np.mean(amps ** 2)))
    pts = torch.tensor(amps / norm, dtype=torch.float64, device=rx.device)
    bt = torch.tensor(bits, device=rx.device)
    out = torch.empty(rx.shape + (Q_m,), dtype=torch.float64, device=rx.device)
    ninf = torch.tensor(float("-inf"), dtype=torch.float64, device=rx.device)
    for rail, y in ((0, rx.real), (1, rx.imag)):
        metric = -((y[..., None] - pts) ** 2) / N0
        for k in range(nb):
            m0 = torch.where(bt[:, k] == 0, metric, ninf)
            m1 = torch.where(bt[:, k] == 1, metric, ninf)
            out[..., 2 * k + rail] = torch.logsumexp(m0, dim=-1) - torch.logsumexp(m1, dim=-1)
    return out.reshape(rx.shape[:-1] + (-1,))


def simulate_point_device(enc_chain, dec_chain, Q_m, EsN0, rv_id_sequence, batch, gen, chan=None):
    """simulate_point with every stage on the GPU (rows N1-N4 of SURVEY.md section 8f): payload RNG, CRC
    attachment, encoding, rate matching, modulation + AWGN + exact LLRs (one HIP kernel, nrldpc_awgn_llr_dev),
    rate recovery, decoding, CRC check, error count.  enc_chain / dec_chain share one NRLDPC parameter object.
    chan: [seed, symbols drawn so far] of the channel's counter-based noise generator (advanced here)."""
    import torch
    from ._capi import awgn_llr_dev
    p = enc_chain.p
    dev = enc_chain.dev
    a = torch.randint(0, 2, (batch, p.A), generator=gen, device=dev, dtype=torch.uint8)   # :118
    N0 = 1.0 / 10.0 ** (EsN0 / 10.0)
    ok = torch.zeros(batch, dtype=torch.bool, device=dev)
    a_hat = torch.zeros((batch, p.A), dtype=torch.uint8, device=dev)
    dec_chain.reset()                                                                      # :122
    for n_rv, rv in enumerate(rv_id_sequence):                                             # :124-137
        p.rv_id = rv
        g = enc_chain.step(a)
        if chan is not None:                                                               # :130-132 in one kernel
            g_tilde = torch.empty(g.shape, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                awgn_llr_dev(g.data_ptr(), g.numel(), Q_m, EsN0, chan[0], chan[1], g_tilde.data_ptr(),
                             torch.cuda.current_stream(dev).cuda_stream)
            chan[1] += g.numel() // Q_m
        else:  # torch elementwise restatement (float64), kept as the cross-check of the kernel
            tx = modulate_t(g, Q_m)
            noise = (N0 / 2.0) ** 0.5 * torch.complex(
                torch.randn(tx.shape, generator=gen, device=dev, dtype=torch.float64),
                torch.randn(tx.shape, generator=gen, device=dev, dtype=torch.float64))
            g_tilde = demodulate_llr_t(tx + noise, Q_m, N0).float()
        dec, good, _ = dec_chain.step(g_tilde)
        newly = good & ~ok
        a_hat = torch.where(newly[:, None], dec, a_hat)  # (no boolean indexing: that is a host synchronisation per batch)
        ok |= good
        if n_rv + 1 < len(rv_id_sequence) and bool(ok.all()):  # the reference stops retransmitting once the block is in (:136)
            break
    return (ok & (a_hat == a).all(dim=1)).cpu().numpy()


def _num2str(x):
    """MATLAB num2str for the values used in the result file name (4 significant decimals, %g-like)."""
    if float(x) == int(x):
        return str(int(x))
    return ("%.5g" % x)


def simulate_point(hEnc, hDec, Q_m, EsN0, rv_id_sequence, batch, rng):
    """One batch of transport blocks at one SNR through the HARQ loop of plot_BLER_vs_SNR.m:118-137.
    Returns per-block success flags (a == a_hat) in simulation order."""
    A = hEnc.A
    a = rng.integers(0, 2, (batch, A), dtype=np.uint8)            # :118
    N0 = 1.0 / 10.0 ** (EsN0 / 10.0)                                 # :106
    ok = np.zeros(batch, bool)
    a_hat = np.zeros((batch, A), np.uint8)
    hDec._nb = batch
    hDec.reset()                                                     # :122
    for rv in rv_id_sequence:                                        # :124-137
        hEnc.rv_id = rv
        hDec.rv_id = rv
        g = hEnc.step_batch(a)                                       # :129
        tx = modulate(g, Q_m)                                        # :130
        noise = np.sqrt(N0 / 2.0) * (rng.standard_normal(tx.shape) + 1j * rng.standard_normal(tx.shape))
        g_tilde = demodulate_llr(tx + noise, Q_m, N0)                # :131-132
        dec, good = hDec.step_batch(g_tilde)                         # :133
        newly = good & ~ok                                           # a block stops retransmitting once decoded
        a_hat[newly] = dec[newly]
        ok |= good
        if ok.all():
            break
    return ok & (a_hat == a).all(axis=1)


def plot_BLER_vs_SNR(A=3842, R=1 / 3, BG=2, Modulation="QPSK", rv_id_sequence=(0,), iterations=8,
                     target_block_errors=3, target_BLER=1e-3, EsN0_start=0.0, EsN0_delta=0.5, seed=0,
                     results_dir="results", batch=256, max_points=200, decoder_kwargs=None, device=False):
    """Same positional parameters and defaults as plot_BLER_vs_SNR.m:1,30-42 (no figure is drawn).
    device=True keeps every stage on the GPU (simulate_point_device).
    Returns {(A, R, BG): [(EsN0, BLER, blocks), ...]}."""
    rng = np.random.default_rng(seed)                                # :45
    Q_m = Q_M.get(Modulation)
    if Q_m is None:
        raise UnsupportedParameters("Unsupported modulation")
    A_list, R_list, BG_list = np.atleast_1d(A), np.atleast_1d(R), np.atleast_1d(BG)
    os.makedirs(results_dir, exist_ok=True)
    curves = {}
    for bg in BG_list:                                               # :53
        for r in R_list:                                             # :54
            for a_len in A_list:                                     # :67
                a_len, bg = int(a_len), int(bg)
                name = "BLER_vs_SNR_%s_%s_%s_%s_%s_%s_%s_%s.txt" % (
                    _num2str(a_len), _num2str(r), _num2str(bg), Modulation, _num2str(iterations),
                    _num2str(target_block_errors), _num2str(EsN0_start), _num2str(seed))  # :79
                points = []
                try:
                    G = int(round(a_len / r / Q_m) * Q_m)            # :94
                    hEnc = NRLDPCEncoder(A=a_len, BG=bg, G=G, Q_m=Q_m)                                   # :98
                    hDec = NRLDPCDecoder(A=a_len, BG=bg, G=G, Q_m=Q_m, I_HARQ=1, iterations=iterations,
                                         **(decoder_kwargs or {}))                                       # :99
                    hEnc.validate()
                    if device:
                        import torch
                        from .device_chain import DeviceDecodeChain, DeviceEncodeChain
                        from .nrldpc import NRLDPC
                        shared = NRLDPC(A=a_len, BG=bg, G=G, Q_m=Q_m)
                        tx_chain = DeviceEncodeChain(shared)
                        rx_chain = DeviceDecodeChain(shared, iterations=iterations, I_HARQ=1, **(decoder_kwargs or {}))
                        gen = torch.Generator(device="cuda")
                        gen.manual_seed(int(seed))
                        chan = [int(seed) * 0x9E3779B97F4A7C15 % (1 << 64), 0]  # noise stream of this curve
                    with open(os.path.join(results_dir, name), "w") as fid:
                        BLER, EsN0, found_start = 1.0, float(EsN0_start), False                          # :84-88
                        while BLER > target_BLER and len(points) < max_points:                           # :104
                            blocks = errors = 0
                            keep_going = True
                            while keep_going and errors < target_block_errors:                           # :116
                                outcomes = (simulate_point_device(tx_chain, rx_chain, Q_m, EsN0, rv_id_sequence, batch, gen, chan)
                                            if device else simulate_point(hEnc, hDec, Q_m, EsN0, rv_id_sequence, batch, rng))
                                for good in outcomes:
                                    if not found_start and not good:                                     # :139-141
                                        keep_going, BLER = False, 1.0
                                        break
                                    found_start = True                                                   # :143
                                    errors += int(not good)                                              # :146-148
                                    blocks += 1                                                          # :152
                                    BLER = errors / blocks                                               # :155
                                    if errors >= target_block_errors:
                                        break
                            if BLER < 1:                                                                 # :164-166
                                fid.write("%f\t%e\n" % (EsN0, BLER))
                                fid.flush()
                                points.append((EsN0, BLER, blocks))
                            EsN0 += EsN0_delta                                                           # :169
                    hEnc.release()
                    hDec.release()
                    if device:
                        tx_chain.close()
                        rx_chain.close()
                except UnsupportedParameters:                        # :172-176: skip this (A, R, BG)
                    continue
                curves[(a_len, float(r), bg)] = points
    return curves
