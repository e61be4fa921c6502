"""NRLDPCEncoder: host-side mirror of the reference's encoder System object (NRLDPCEncoder.m).

step(a) runs the six TS 38.212 stages of NRLDPCEncoder.m:60-67; stage 3 (LDPC coding, :127-165)
is the GPU encoder behind the C ABI (nrldpc_encode replaces step(obj.hLDPCEncoder, c) at :158).
step_batch(a[B][A]) is the same chain for B transport blocks at once (all code blocks of all
transport blocks in one launch) -- the reference encodes one block per step().
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
        a = np.asarray(a)
        if a.ndim == 2 and a.shape[1] == 1:
            a = a[:, 0]
        if a.ndim != 1 or a.size != self.A:
            raise NRLDPCError("a should be a column vector of length A.")
        return self.step_batch(a[None, :])[0]

    def step_batch(self, a):
        """a: [B][A] bits -> g: [B][G] bits."""
        if self._codec is None:
            self._setup()
        else:
            self.validate()
        a = np.asarray(a)
        if a.ndim != 2 or a.shape[1] != self.A:
            raise NRLDPCError("a should be a column vector of length A.")
        b = self.crc_calculation(a.astype(np.uint8))
        c = self.code_block_segmentation(b)
        d = self.LDPC_coding(c)
        return self.rate_match(d)

    # -- stages (all on [B][...] arrays) -----------------------------------------------------------
    def crc_calculation(self, a):  # NRLDPCEncoder.m:70-89
        poly, L = self.transport_block_CRC_polynomial, self.transport_block_L
        return np.concatenate([a, chain.crc_bits(a, poly, L)], axis=1)

    def code_block_segmentation(self, b):  # NRLDPCEncoder.m:92-124; filler bits (NaN there) are encoded as 0 (:153)
        C_, K_, Kp, L = self.C, self.K, int(self.K_prime), self.code_block_L
        c = np.zeros((b.shape[0], C_, K_), np.uint8)
        s = 0
        for r in range(C_):
            c[:, r, : Kp - L] = b[:, s: s + Kp - L]
            s += Kp - L
            if C_ > 1:
                c[:, r, Kp - L: Kp] = chain.crc_bits(c[:, r, : Kp - L], self.code_block_CRC_polynomial, L)
        return c

    def LDPC_coding(self, c):  # NRLDPCEncoder.m:127-165
        B, C_, K_ = c.shape
        cw = self._codec.encode(c.reshape(B * C_, K_)).reshape(B, C_, -1)  # systematic [c; w]
        return cw[:, :, 2 * self.Z_c:]  # d = cw without the 2Z punctured columns (:149-163)

    def rate_match(self, d, _filler=None):  # bit selection + interleaving + concatenation (:168-256)
        d = np.asarray(d)
        single = d.ndim == 2
        if single:
            d = d[None]
        g = np.zeros((d.shape[0], self.G), np.uint8)
        for r, (off, dpos, fpos) in enumerate(chain.g_to_d_maps(self)):
            if dpos.size:
                g[:, off + fpos] = d[:, r, dpos]
        return g[0] if single else g
