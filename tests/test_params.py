"""Parameter chain of NRLDPC.m:297-543 -- known answers from SURVEY.md section 8 -- and the
reference's validation behaviour (NRLDPC.m:240-294, 551-559)."""
import pytest

KNOWN = [  # kwargs, expected
    (dict(BG=2, A=100, G=300, Q_m=2), dict(L=16, B=116, C=1, Kp=116, Kb=6, Z=20, iLS=2, K=200, N=1000, E=[300], nl=12)),
    (dict(BG=1, A=8424, G=25272, Q_m=2), dict(L=24, B=8448, C=1, Kp=8448, Kb=22, Z=384, iLS=1, K=8448, N=25344, E=[25272], nl=46)),
    (dict(BG=2, A=3824, G=19120, Q_m=2), dict(L=16, B=3840, C=1, Kp=3840, Kb=10, Z=384, iLS=1, K=3840, N=19200, E=[19120], nl=42)),
    (dict(BG=2, A=3842, G=11526, Q_m=2), dict(L=24, B=3866, C=2, Kp=1957, Kb=10, Z=208, iLS=6, K=2080, N=10400, E=[5762, 5764], nl=21)),
    (dict(BG=1, A=8424, G=9478, Q_m=2), dict(L=24, B=8448, C=1, Kp=8448, Kb=22, Z=384, iLS=1, K=8448, N=25344, E=[9478], nl=5)),
]


@pytest.mark.parametrize("kw,exp", KNOWN)
def test_known_answers(pkg, kw, exp):
    p = pkg.NRLDPC(**kw)
    p.validate()
    got = dict(L=p.transport_block_L, B=p.B, C=p.C, Kp=p.K_prime, Kb=p.K_b, Z=p.Z_c, iLS=p.i_LS, K=p.K, N=p.N,
               E=p.E_r, nl=p.active_layers())
    assert got == exp


def test_cfg3_rate_sweep_active_layers(pkg):
    """BG2 Z=384 rate sweep of BASELINE.json configs[2]: G -> active layers."""
    for G, layers in zip((19120, 15296, 11472, 9560, 7648, 6374, 5736), (42, 32, 22, 17, 12, 9, 7)):
        assert pkg.NRLDPC(BG=2, A=3824, G=G, Q_m=2).active_layers() == layers


def test_k0_table(pkg):
    p = pkg.NRLDPC(BG=1, A=8424, G=25272, Q_m=2)
    ks = []
    for rv in range(4):
        p.rv_id = rv
        ks.append(p.k_0)
    assert ks == [0, 17 * 384, 33 * 384, 56 * 384]
    q = pkg.NRLDPC(BG=2, A=3824, G=19120, Q_m=2, I_LBRM=1, TBS_LBRM=5000)
    assert q.N_ref == 7500 and q.N_cb == 7500
    q.rv_id = 2
    assert q.k_0 == (25 * 7500) // (50 * 384) * 384


def test_setters_reject_like_the_reference(pkg):
    U = pkg.UnsupportedParameters
    for kw in (dict(BG=3), dict(A=-1), dict(rv_id=4), dict(Q_m=3), dict(N_L=5), dict(G=-2), dict(TBS_LBRM=-1)):
        with pytest.raises(U):
            pkg.NRLDPC(**kw)
    with pytest.raises(U):
        pkg.NRLDPC(BG=1, A=100, G=301, Q_m=2).validate()  # G not a multiple of Q_m*N_L
    with pytest.raises(U):
        pkg.NRLDPC(BG=1, A=8425, G=30000).validate()  # B' not a multiple of C (NRLDPC.m:552-554)
    assert U.identifier == "ldpc_3gpp_matlab:UnsupportedParameters"
    assert pkg.NRLDPCError.identifier == "ldpc_3gpp_matlab:Error"


def test_er_split_and_cbgti(pkg):
    p = pkg.NRLDPC(BG=1, A=20016, G=60000, Q_m=4, N_L=2)
    assert p.C == 3 and sum(p.E_r) == 60000 and all(e % 8 == 0 for e in p.E_r)
    p.CBGTI = [1]
    assert p.CBGTI_flags == [1, 0, 1] and p.E_r[1] == 0 and sum(p.E_r) == 60000


def test_crc_polynomial_lookup(pkg):
    assert pkg.get_3gpp_crc_polynomial("CRC24A") == (0x1864CFB, 24)
    assert pkg.get_3gpp_crc_polynomial("CRC24B") == (0x1800063, 24)
    assert pkg.get_3gpp_crc_polynomial("CRC16") == (0x11021, 16)
    assert pkg.get_3gpp_crc_polynomial("None") == (0, 0)
    with pytest.raises(pkg.UnsupportedParameters):
        pkg.get_3gpp_crc_polynomial("CRC8")
