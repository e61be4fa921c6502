"""NRLDPCEncoder: host-side mirror of the reference's encoder System object (NRLDPCEncoder.m).

step(a) runs the six TS 38.212 stages of NRLDPCEncoder.m:60-67; stage 3 (LDPC coding, :127-165)
is the GPU encoder behind the C ABI (nrldpc_encode replaces step(obj.hLDPCEncoder, c) at :158).
"""
import numpy as np

from . import chain
from ._capi import Codec, NRLDPCError
from .nrldpc import NRLDPC


class NRLDPCEncoder(NRLDPC):
    def __init__(self, device_id=0, **kw):
        super().__init__(**kw)
        self._device_id = device_id
        self._codec = None

    # -- System-object protocol (NRLDPCEncoder.m:44-50) -------------------------------------------
    def _setup(self):
        self.validate()
        self._codec = Codec(self.BG, self.Z_c, max_iter=1, llr_dtype=np.float32, device_id=self._device_id)
        object.__setattr__(self, "_locked", True)

    def release(self):
        if self._codec is not None:
            self._codec.close()
            self._codec = None
        super().release()

    def reset(self):  # NRLDPCEncoder.m:266-268: nothing to reset
        pass

    def __call__(self, a):
        return self.step(a)

    def step(self, a):
        """a: A information bits (0/1) -> g: G encoded bits (NRLDPCEncoder.m:60-67)."""
        if self._codec is None:
            self._setup()
        else:
            self.validate()
        a = np.asarray(a)
        if a.ndim == 2 and a.shape[1] == 1:
            a = a[:, 0]
        if a.ndim != 1 or a.size != self.A:
            raise NRLDPCError("a should be a column vector of length A.")
        b = self.crc_calculation(a.astype(np.uint8))
        c, filler = self.code_block_segmentation(b)
        d = self.LDPC_coding(c)
        return self.rate_match(d, filler)

    # -- stages ----------------------------------------------------------------------------------
    def crc_calculation(self, a):  # NRLDPCEncoder.m:70-89
        poly, L = self.transport_block_CRC_polynomial, self.transport_block_L
        return np.concatenate([a, chain.crc_bits(a, poly, L)])

    def code_block_segmentation(self, b):  # NRLDPCEncoder.m:92-124
        C_, K_, Kp, L = self.C, self.K, int(self.K_prime), self.code_block_L
        c = np.zeros((C_, K_), np.uint8)
        s = 0
        for r in range(C_):
            c[r, : Kp - L] = b[s: s + Kp - L]
            s += Kp - L
            if C_ > 1:
                c[r, Kp - L: Kp] = chain.crc_bits(c[r, : Kp - L], self.code_block_CRC_polynomial, L)
        filler = np.zeros(K_, bool)
        filler[Kp:] = True  # <NULL> bits, NaN in the reference (:120-122); encoded as 0 (:153)
        return c, filler

    def LDPC_coding(self, c):  # NRLDPCEncoder.m:127-165
        cw = self._codec.encode(c)  # [C][N + 2Z], systematic
        return cw[:, 2 * self.Z_c:]  # d = cw without the 2Z punctured columns (:149-163)

    def rate_match(self, d, filler):  # bit selection + interleaving + concatenation (:168-256)
        g = np.zeros(self.G, np.uint8)
        for r, (off, dpos, fpos) in enumerate(chain.g_to_d_maps(self)):
            if dpos.size:
                g[off + fpos] = d[r, dpos]
        return g
