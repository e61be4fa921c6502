"""NRLDPCEncoder / NRLDPCDecoder System-object mirrors (the API plot_BLER_vs_SNR.m:98-133 and
testbench.m:39-49 drive): step(), reset(), tunable rv_id, HARQ, CRC failure -> empty output."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def qpsk_awgn_llr(rng, g, esn0_db):
    mu = 2.0 * 10.0 ** (esn0_db / 10.0)
    return (1 - 2.0 * g) * mu + np.sqrt(2 * mu) * rng.standard_normal(g.size)


@pytest.mark.parametrize("kw,esn0", [
    (dict(BG=2, A=100, G=300, Q_m=2), 5.0),                 # BASELINE configs[0]
    (dict(BG=2, A=3842, G=11526, Q_m=2), 1.5),              # plot_BLER_vs_SNR.m defaults: C = 2, CB-CRC path
    (dict(BG=1, A=8424, G=25272, Q_m=2), 0.0),              # headline code
    (dict(BG=1, A=20016, G=60000, Q_m=4, N_L=2), 1.0),      # C = 3
    (dict(BG=2, A=500, G=5004, Q_m=6, rv_id=0), 0.0),       # heavy repetition (soft combining)
    (dict(BG=1, A=5000, G=7000, Q_m=2, I_LBRM=1, TBS_LBRM=9000), 5.0)])  # limited buffer
def test_step_round_trip(pkg, kw, esn0):
    rng = np.random.default_rng(kw["A"])
    enc = pkg.NRLDPCEncoder(**kw)
    dec = pkg.NRLDPCDecoder(iterations=25, **kw)
    ok = 0
    for _ in range(4):
        a = rng.integers(0, 2, kw["A"], dtype=np.uint8)
        g = enc(a)                                   # functor form as in testbench.m:49
        assert g.shape == (kw["G"],) and set(np.unique(g)) <= {0, 1}
        a_hat = dec.step(qpsk_awgn_llr(rng, g, esn0))
        assert a_hat.size in (0, kw["A"])
        ok += int(a_hat.size == kw["A"] and (a_hat == a).all())
    assert ok >= 3
    enc.release()
    dec.release()


def test_crc_failure_returns_empty(pkg):
    kw = dict(BG=2, A=3842, G=11526, Q_m=2)
    rng = np.random.default_rng(1)
    enc, dec = pkg.NRLDPCEncoder(**kw), pkg.NRLDPCDecoder(iterations=5, **kw)
    g = enc.step(rng.integers(0, 2, kw["A"], dtype=np.uint8))
    a_hat = dec.step(qpsk_awgn_llr(rng, g, -8.0))  # hopeless SNR
    assert a_hat.size == 0
    with pytest.raises(pkg.NRLDPCError):
        dec.step(np.zeros(kw["G"] + 1))
    with pytest.raises(pkg.NRLDPCError):
        enc.step(np.zeros(kw["A"] - 1))


def test_harq_incremental_redundancy(pkg):
    """plot_BLER_vs_SNR.m:120-137: I_HARQ=1, reset per block, rv_id sequence until a_hat is non-empty."""
    kw = dict(BG=1, A=4000, G=4800, Q_m=2)  # R ~ 0.83 per transmission
    rng = np.random.default_rng(2)
    enc = pkg.NRLDPCEncoder(**kw)
    dec = pkg.NRLDPCDecoder(I_HARQ=1, iterations=25, **kw)
    first_try = combined = 0
    for _ in range(6):
        a = rng.integers(0, 2, kw["A"], dtype=np.uint8)
        dec.reset()
        a_hat = np.zeros(0)
        for n, rv in enumerate((0, 2, 3, 1)):
            enc.rv_id = rv
            dec.rv_id = rv
            a_hat = dec.step(qpsk_awgn_llr(rng, enc.step(a), 2.0))
            if a_hat.size:
                break
        assert a_hat.size and (a_hat == a).all()
        first_try += int(n == 0)
        combined += int(n > 0)
    assert combined >= 3  # at 2 dB a single R=0.83 transmission usually fails, combining succeeds


def test_decoder_matches_unpruned_reference_semantics(pkg):
    """Row pruning (active layers only) must not change hard decisions vs decoding the full H, which is
    what the reference always does (NRLDPCDecoder.m:120)."""
    kw = dict(BG=2, A=3824, G=7648, Q_m=2)
    rng = np.random.default_rng(3)
    enc = pkg.NRLDPCEncoder(**kw)
    d1 = pkg.NRLDPCDecoder(iterations=20, alpha=0.75, **kw)
    d2 = pkg.NRLDPCDecoder(iterations=20, alpha=0.75, prune_layers=False, **kw)
    for _ in range(3):
        a = rng.integers(0, 2, kw["A"], dtype=np.uint8)
        y = qpsk_awgn_llr(rng, enc.step(a), 2.5)
        r1, r2 = d1.step(y), d2.step(y)
        assert r1.size == r2.size and (r1 == r2).all() and (r1 == a).all()
