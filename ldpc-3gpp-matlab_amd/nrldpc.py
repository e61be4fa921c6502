"""Host-side mirror of the reference's NRLDPC base System object (NRLDPC.m).

Same property names, the same derivation chain (NRLDPC.m:297-543) and the same validation /
error behaviour (NRLDPC.m:240-294, 551-559): invalid settings raise UnsupportedParameters, which
callers such as plot_BLER_vs_SNR.m:172-176 and testbench.m:48-56 catch and skip.  Like the
reference's Dependent properties, everything is re-derived from the settable properties on access.
This is host logic (integer arithmetic on a handful of scalars); the data path lives in the HIP
library behind _capi.Codec.
"""
import math

from ._capi import UnsupportedParameters, lifting_size, set_index

# get_3gpp_crc_polynomial.m:3-14 -- generator polynomials as integers (MSB = highest power)
CRC_POLY = {
    "CRC24A": (0x1864CFB, 24),  # z^24+z^23+z^18+z^17+z^14+z^11+z^10+z^7+z^6+z^5+z^4+z^3+z+1
    "CRC24B": (0x1800063, 24),  # z^24+z^23+z^6+z^5+z+1
    "CRC16": (0x11021, 16),     # z^16+z^12+z^5+1
    "None": (0, 0),
}


def get_3gpp_crc_polynomial(crc):
    """(polynomial as int, L) for 'CRC24A' | 'CRC24B' | 'CRC16' | 'None' (get_3gpp_crc_polynomial.m:1-19)."""
    try:
        return CRC_POLY[crc]
    except KeyError:
        raise UnsupportedParameters("Invalid CRC identifier.")


class NRLDPC:
    """Parameter object: Nontunable BG, A, I_LBRM, TBS_LBRM; tunable rv_id, G, Q_m, N_L, CBGTI
    (NRLDPC.m:20-85).  Construct with keyword arguments, e.g. NRLDPC(BG=1, A=20, G=132)."""

    _NONTUNABLE = ("BG", "A", "I_LBRM", "TBS_LBRM")
    _TUNABLE = ("rv_id", "G", "Q_m", "N_L", "CBGTI")

    def __init__(self, **kw):
        object.__setattr__(self, "_locked", False)
        self._BG, self._A, self._I_LBRM, self._TBS_LBRM = 1, 44, 0, math.inf  # NRLDPC.m:28-46
        self._rv_id, self._G, self._Q_m, self._N_L, self._CBGTI = 0, 132, 1, 1, []  # NRLDPC.m:57-84
        for k, v in kw.items():
            if k not in self._settable():
                raise AttributeError("unknown property %r" % k)
            setattr(self, k, v)

    @classmethod
    def _settable(cls):
        return cls._NONTUNABLE + cls._TUNABLE

    # -- settable properties with the reference's range checks (NRLDPC.m:240-294) -----------------
    def _set_nontunable(self, name, value):
        if self._locked:
            raise RuntimeError("%s is Nontunable: call release() before changing it" % name)
        object.__setattr__(self, "_" + name, value)

    BG = property(lambda s: s._BG)

    @BG.setter
    def BG(self, v):
        if v < 1 or v > 2:
            raise UnsupportedParameters("Valid values of BG are 1 and 2.")
        self._set_nontunable("BG", int(v))

    A = property(lambda s: s._A)

    @A.setter
    def A(self, v):
        if v < 0:
            raise UnsupportedParameters("A should not be negative.")
        self._set_nontunable("A", int(v))

    I_LBRM = property(lambda s: s._I_LBRM)

    @I_LBRM.setter
    def I_LBRM(self, v):
        self._set_nontunable("I_LBRM", int(v))

    TBS_LBRM = property(lambda s: s._TBS_LBRM)

    @TBS_LBRM.setter
    def TBS_LBRM(self, v):
        if v < 0:
            raise UnsupportedParameters("TBS_LBRM should not be negative.")
        self._set_nontunable("TBS_LBRM", v)

    rv_id = property(lambda s: s._rv_id)

    @rv_id.setter
    def rv_id(self, v):
        if v < 0 or v > 3:
            raise UnsupportedParameters("Valid values of rv_id are 0, 1, 2 and 3.")
        self._rv_id = int(v)

    G = property(lambda s: s._G)

    @G.setter
    def G(self, v):
        if v < 0:
            raise UnsupportedParameters("G should not be negative.")
        self._G = int(v)

    Q_m = property(lambda s: s._Q_m)

    @Q_m.setter
    def Q_m(self, v):
        if v not in (1, 2, 4, 6, 8):
            raise UnsupportedParameters("Valid vales of Q_m are 1, 2, 4, 6 and 8.")
        self._Q_m = int(v)

    N_L = property(lambda s: s._N_L)

    @N_L.setter
    def N_L(self, v):
        if v < 1 or v > 4:
            raise UnsupportedParameters("N_L should be in the range 1 to 4.")
        self._N_L = int(v)

    CBGTI = property(lambda s: s._CBGTI)

    @CBGTI.setter
    def CBGTI(self, v):
        self._CBGTI = [int(x) for x in v]

    # -- dependent properties (NRLDPC.m:297-543) ---------------------------------------------------
    @property
    def transport_block_CRC(self):  # NRLDPC.m:297-303
        return "CRC24A" if self.A > 3824 else "CRC16"

    @property
    def transport_block_CRC_polynomial(self):
        return get_3gpp_crc_polynomial(self.transport_block_CRC)[0]

    @property
    def transport_block_L(self):
        return get_3gpp_crc_polynomial(self.transport_block_CRC)[1]

    @property
    def B(self):  # NRLDPC.m:316-318
        return self.A + self.transport_block_L

    @property
    def K_cb(self):  # NRLDPC.m:321-331
        return 8448 if self.BG == 1 else 3840

    @property
    def code_block_CRC(self):  # NRLDPC.m:347-353
        return "None" if self.B <= self.K_cb else "CRC24B"

    @property
    def code_block_CRC_polynomial(self):
        return get_3gpp_crc_polynomial(self.code_block_CRC)[0]

    @property
    def code_block_L(self):
        return get_3gpp_crc_polynomial(self.code_block_CRC)[1]

    def code_block_check(self):
        """(generator polynomial with its x^L term, L, K') of the CRC that closes every code block: the code-block CRC24B
        when the transport block is segmented, else the transport block's own CRC (NRLDPCDecoder.m:298-301, 336) -- the
        `crc` argument of Codec for the CRC-aided stop (nrldpc_cfg.early_term = 2)."""
        if self.C > 1:
            return (self.code_block_CRC_polynomial, self.code_block_L, int(self.K_prime))
        return (self.transport_block_CRC_polynomial, self.transport_block_L, int(self.K_prime))

    @property
    def C(self):  # NRLDPC.m:334-344
        if self.B <= self.K_cb:
            return 1
        return -(-self.B // (self.K_cb - self.code_block_L))

    @property
    def B_prime(self):  # NRLDPC.m:366-377
        return self.B if self.B <= self.K_cb else self.B + self.C * self.code_block_L

    @property
    def K_prime(self):  # NRLDPC.m:380-382 (a non-integer value is rejected by validate())
        bp, c = self.B_prime, self.C
        return bp // c if bp % c == 0 else bp / c

    @property
    def K_b(self):  # NRLDPC.m:385-406
        if self.BG == 1:
            return 22
        kp = self.K_prime
        if kp > 640:
            return 10
        if kp > 560:
            return 9
        if kp > 192:
            return 8
        return 6

    @property
    def Z_c(self):  # NRLDPC.m:409-411 -> get_3gpp_lifting_size.m
        z = lifting_size(self.K_b, int(math.ceil(self.K_prime)))
        if z < 0:
            raise UnsupportedParameters("Invalid block length.")
        return z

    @property
    def K(self):  # NRLDPC.m:414-425
        return self.Z_c * (22 if self.BG == 1 else 10)

    @property
    def i_LS(self):  # NRLDPC.m:428-430 -> get_3gpp_set_index.m
        i = set_index(self.Z_c)
        if i < 0:
            raise UnsupportedParameters("Invalid lifting size.")
        return i

    @property
    def N(self):  # NRLDPC.m:443-454
        return self.Z_c * (66 if self.BG == 1 else 50)

    @property
    def N_ref(self):  # NRLDPC.m:457-460, R_LBRM = 2/3
        if math.isinf(self.TBS_LBRM):
            return math.inf
        return math.floor(self.TBS_LBRM / (self.C * (2.0 / 3.0)))

    @property
    def N_cb(self):  # NRLDPC.m:463-469
        if self.I_LBRM == 0:
            return self.N
        return int(min(self.N, self.N_ref))

    @property
    def CBGTI_flags(self):  # NRLDPC.m:471-477: 1 = code block is (re)transmitted
        flags = [1] * self.C
        for r in self.CBGTI:
            if r < self.C:
                flags[r] = 0
        return flags

    @property
    def C_prime(self):  # NRLDPC.m:480-482
        return sum(self.CBGTI_flags)

    @property
    def E_r(self):  # NRLDPC.m:485-507
        C_, Cp, flags = self.C, self.C_prime, self.CBGTI_flags
        G_, NL, Qm = self.G, self.N_L, self.Q_m
        out, j = [0] * C_, 0
        for r in range(C_):
            if flags[r] == 0:
                continue
            per = G_ / (NL * Qm)
            if j <= Cp - (per % Cp) - 1:
                out[r] = NL * Qm * math.floor(G_ / (NL * Qm * Cp))
            else:
                out[r] = NL * Qm * math.ceil(G_ / (NL * Qm * Cp))
            j += 1
        return [int(x) for x in out]

    @property
    def k_0(self):  # NRLDPC.m:510-543, Table 5.4.2.1-2
        N_cb, Z = self.N_cb, self.Z_c
        num = {1: (0, 17, 33, 56), 2: (0, 13, 25, 43)}[self.BG][self.rv_id]
        den = 66 if self.BG == 1 else 50
        return (num * N_cb) // (den * Z) * Z

    # -- System-object protocol -------------------------------------------------------------------
    def validate(self):
        """validatePropertiesImpl (NRLDPC.m:551-559)."""
        if self.B_prime % self.C != 0:
            raise UnsupportedParameters("B_prime must be a multiple of C.")
        if self.G % (self.Q_m * self.N_L) != 0:
            raise UnsupportedParameters("G must be a multiple of Q_m*N_L.")

    def release(self):
        object.__setattr__(self, "_locked", False)

    def active_layers(self):
        """Base-graph rows that can carry a non-zero channel LLR for the current (rv_id, E_r, N_cb):
        rows whose extension-parity column lies at or below the highest circular-buffer position any
        code block receives.  Rows above it have an all-zero-LLR degree-1 column, so their messages are
        identically zero; the reference always decodes the full H (NRLDPCDecoder.m:120).
        """
        import numpy as np
        Z, N_cb, k0 = self.Z_c, self.N_cb, self.k_0
        nrows = 46 if self.BG == 1 else 42
        kb_cols = 22 if self.BG == 1 else 10
        E = max(self.E_r) if self.E_r else 0
        if E <= 0:
            return 4
        # circular-buffer walk of NRLDPCDecoder.m:226-234: positions (k0+j) mod N_cb, fillers skipped
        pos = (k0 + np.arange(N_cb)) % N_cb
        lo_f, hi_f = max(int(self.K_prime) - 2 * Z, 0), self.K - 2 * Z
        pos = pos[(pos < lo_f) | (pos >= hi_f)]
        hi = int(pos[: min(E, pos.size)].max()) + 1
        top_col = (hi - 1 + 2 * Z) // Z  # base column of the highest d position that receives an LLR
        layers = top_col - kb_cols + 1
        return int(min(max(layers, 4), nrows))
