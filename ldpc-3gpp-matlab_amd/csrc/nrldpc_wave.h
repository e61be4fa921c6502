// nrldpc_wave.h -- wave-local LDS staging helpers shared by the CRC and encoder kernels (one wave64 owns a
// code block; no workgroup barriers).
#ifndef NRLDPC_WAVE_H
#define NRLDPC_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrldpc {

// LDS written by some lanes of a wave is read by other lanes of the same wave: DS instructions of one wave
// execute in issue order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Copy n bytes global -> LDS by one wave with 16-byte loads whatever the source alignment: the LDS copy is
// placed at the same offset modulo 16 as the source (`base` is 16-byte aligned with 16 bytes of slack), so
// only the first and last few bytes move one at a time.  Returns the LDS address of byte 0.
__device__ __forceinline__ uint8_t* stage_row(uint8_t* base, const uint8_t* src, int n) {
    const int lane = threadIdx.x & 63;
    const int sh = (int)(reinterpret_cast<uintptr_t>(src) & 15);
    uint8_t* dst = base + sh;
    int head = (16 - sh) & 15;
    head = head < n ? head : n;
    if (lane < head) dst[lane] = src[lane];
    const int nv = (n - head) >> 4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src + head);
    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
    // four 16-byte loads in flight per lane before the first LDS store: the stages are latency-bound at one wave per
    // code block, and a load per loop trip left ~1 KB per wave on the wire
    int i = lane;
    for (; i + 192 < nv; i += 256) {
        const uint4 a = s4[i], b = s4[i + 64], c = s4[i + 128], d = s4[i + 192];
        d4[i] = a; d4[i + 64] = b; d4[i + 128] = c; d4[i + 192] = d;
    }
    for (; i < nv; i += 64) d4[i] = s4[i];
    const int done = head + (nv << 4);
    if (lane < n - done) dst[done + lane] = src[done + lane];
    return dst;
}

// Copy n bytes LDS -> global by one wave with 16-byte stores whatever the two alignments (an unaligned LDS
// dword is two aligned reads + v_alignbyte; up to 4 bytes past the end are read, never stored), masking
// every byte to its bit.
__device__ __forceinline__ void store_row(uint8_t* dst, const uint8_t* src, int n) {
    const int lane = threadIdx.x & 63;
    int head = (16 - (int)(reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    head = head < n ? head : n;
    if (lane < head) dst[lane] = src[lane] & 1u;
    const int nv = (n - head) >> 4;
    const uintptr_t so = reinterpret_cast<uintptr_t>(src + head);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(so & ~(uintptr_t)3);
    const uint32_t rot = (uint32_t)(so & 3);
    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
    for (int i = lane; i < nv; i += 64) {
        const uint32_t* q = s32 + 4 * i;
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
        uint4 o;
        o.x = __builtin_amdgcn_alignbyte(w1, w0, rot) & 0x01010101u;
        o.y = __builtin_amdgcn_alignbyte(w2, w1, rot) & 0x01010101u;
        o.z = __builtin_amdgcn_alignbyte(w3, w2, rot) & 0x01010101u;
        o.w = __builtin_amdgcn_alignbyte(w4, w3, rot) & 0x01010101u;
        d4[i] = o;
    }
    const int done = head + (nv << 4);
    if (lane < n - done) dst[done + lane] = src[done + lane] & 1u;
}

} // namespace nrldpc
#endif
