"""testbench.m as data: the random parameter draws of /root/reference/testbench.m:21-36 and a SECOND, function-style
transcription of the parameter chain NRLDPC.m:297-543 + validatePropertiesImpl (:551-559) + the lifting-size / set-index
helpers -- written straight from the .m text, sharing no code with ldpc-3gpp-matlab_amd/nrldpc.py (the class-style mirror
the product uses), so that an error in either shows up as a disagreement.  Test infrastructure only."""
import math

import numpy as np


class Unsupported(Exception):
    """'ldpc_3gpp_matlab:UnsupportedParameters' (testbench.m:48-56 catches it and draws again)."""


def draw(rng):
    """testbench.m:21-36, statement by statement (rand -> rng.random(), randi(n) -> 1 + integers(n))."""
    R = rng.random()
    I_LBRM = int(round(rng.random()))
    A = int(math.ceil(100000 ** rng.random()))
    TBS_LBRM = int(round(A / max(rng.random(), 1e-12)))
    if A <= 292 or (A <= 3824 and R <= 0.67) or R <= 0.25:  # :26-30
        BG = 2
    else:
        BG = 1
    Q_m = (1, 2, 4, 6, 8)[int(rng.integers(0, 5))]
    N_L = 1 + int(rng.integers(0, 4))
    G = Q_m * N_L * int(round(A / max(rng.random(), 1e-12) / Q_m / N_L))
    rv_id = int(rng.integers(0, 4))
    return dict(BG=BG, A=A, I_LBRM=I_LBRM, TBS_LBRM=TBS_LBRM, rv_id=rv_id, G=G, Q_m=Q_m, N_L=N_L)


# get_3gpp_valid_lifting_sizes.m / get_3gpp_set_index.m / get_3gpp_lifting_size.m (Table 5.3.2-1 of TS 38.212)
_SETS = [[2, 4, 8, 16, 32, 64, 128, 256], [3, 6, 12, 24, 48, 96, 192, 384], [5, 10, 20, 40, 80, 160, 320],
         [7, 14, 28, 56, 112, 224], [9, 18, 36, 72, 144, 288], [11, 22, 44, 88, 176, 352], [13, 26, 52, 104, 208],
         [15, 30, 60, 120, 240]]


def _lifting_size(K_b, K_prime):
    ok = [z for s in _SETS for z in s if K_b * z >= K_prime]
    if not ok:
        raise Unsupported("Invalid block length.")
    return min(ok)


def _set_index(Z):
    for i, s in enumerate(_SETS):
        if Z in s:
            return i
    raise Unsupported("Invalid lifting size.")


def derive(BG, A, I_LBRM, TBS_LBRM, rv_id, G, Q_m, N_L, CBGTI=()):
    """Every Dependent property of NRLDPC.m, in the order the file defines them, then validatePropertiesImpl."""
    L_tb = 24 if A > 3824 else 16                                   # :297-313
    B = A + L_tb                                                    # :316-318
    K_cb = 8448 if BG == 1 else 3840                                # :321-331
    L_cb = 0 if B <= K_cb else 24                                   # :347-363
    C = 1 if B <= K_cb else math.ceil(B / (K_cb - L_cb))            # :334-344
    B_prime = B if B <= K_cb else B + C * L_cb                      # :366-377
    if B_prime % C != 0:                                            # :552-554
        raise Unsupported("B_prime must be a multiple of C.")
    if G % (Q_m * N_L) != 0:                                        # :556-558
        raise Unsupported("G must be a multiple of Q_m*N_L.")
    K_prime = B_prime // C                                          # :380-382
    if BG == 1:                                                     # :385-406
        K_b = 22
    elif K_prime > 640:
        K_b = 10
    elif K_prime > 560:
        K_b = 9
    elif K_prime > 192:
        K_b = 8
    else:
        K_b = 6
    Z_c = _lifting_size(K_b, K_prime)                               # :409-411
    K = Z_c * (22 if BG == 1 else 10)                               # :414-425
    i_LS = _set_index(Z_c)                                          # :428-430
    N = Z_c * (66 if BG == 1 else 50)                               # :443-454
    N_ref = math.floor(TBS_LBRM / (C * (2.0 / 3.0)))                # :457-460
    N_cb = N if I_LBRM == 0 else min(N, N_ref)                      # :463-469
    flags = [1] * C                                                 # :471-477
    for r in CBGTI:
        if r < C:
            flags[r] = 0
    C_prime = sum(flags)                                            # :480-482
    E_r, j = [0] * C, 0                                             # :485-507
    for r in range(C):
        if flags[r]:
            if j <= C_prime - ((G // (N_L * Q_m)) % C_prime) - 1:
                E_r[r] = N_L * Q_m * (G // (N_L * Q_m * C_prime))
            else:
                E_r[r] = N_L * Q_m * -(-G // (N_L * Q_m * C_prime))
            j += 1
    num = {1: (0, 17, 33, 56), 2: (0, 13, 25, 43)}[BG][rv_id]       # :510-543
    k_0 = math.floor((num * N_cb) / ((66 if BG == 1 else 50) * Z_c)) * Z_c
    return dict(transport_block_L=L_tb, B=B, K_cb=K_cb, code_block_L=L_cb, C=C, B_prime=B_prime, K_prime=K_prime, K_b=K_b,
                Z_c=Z_c, K=K, i_LS=i_LS, N=N, N_ref=N_ref, N_cb=N_cb, C_prime=C_prime, E_r=E_r, k_0=k_0)


def literal_encoder_tail(d, p):
    """NRLDPCEncoder.m:168-256 for one transport block, per-element loops as the reference writes them: bit selection
    (:186-195, NaN skip), bit interleaving (:219-223), code block concatenation (:243-253).  d: [C][N] uint8 with 2 =
    NaN (filler); p: dict from derive() plus G, Q_m."""
    out = []
    for r in range(p["C"]):
        E = p["E_r"][r]
        e, k, j = np.zeros(E, np.uint8), 0, 0
        while k < E:
            x = d[r][(p["k_0"] + j) % p["N_cb"]]
            if x != 2:
                e[k] = x
                k += 1
            j += 1
        f = np.zeros(E, np.uint8)
        Q = p["Q_m"]
        for jj in range(E // Q):
            for i in range(Q):
                f[i + jj * Q] = e[i * (E // Q) + jj]
        out.append(f)
    return np.concatenate(out) if out else np.zeros(0, np.uint8)
