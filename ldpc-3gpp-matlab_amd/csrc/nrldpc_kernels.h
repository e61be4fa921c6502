// nrldpc_kernels.h -- argument blocks and launchers shared by the C ABI and the HIP kernels.
#ifndef NRLDPC_KERNELS_H
#define NRLDPC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NRLDPC_K_F32 0
#define NRLDPC_K_F16 1

#ifndef NRLDPC_GEN_THREADS_BG1
#define NRLDPC_GEN_THREADS_BG1 512 // workgroup size cap of the run-time-Z kernel for BG1
#endif
#ifndef NRLDPC_GEN_THREADS_BG2
#define NRLDPC_GEN_THREADS_BG2 512 // same for BG2 (register allocation: 6 waves per SIMD)
#endif
#ifndef NRLDPC_GEN_WPE_BG2
#define NRLDPC_GEN_WPE_BG2 6
#endif
#ifndef NRLDPC_GEN_WPE_BG1
#define NRLDPC_GEN_WPE_BG1 4     // waves per SIMD its register allocation is sized for
#endif

namespace nrldpc {

constexpr int CRC_SLOTS = 20; // LDS ints per codeword of the CRC-aided stop: 16 partial remainders, 1 "a bit is set" flag, 3 of padding

struct DecArgs {
    const void* llr;     // [batch][ncols*Z] f32 or f16
    uint8_t* hard;       // [batch][kb*Z]
    int32_t* iters;      // [batch] or null
    float* app;          // [batch][ncols*Z] or null
    const int32_t* rot;  // per table edge: ring rotation P_e * sbw in bytes (nrldpc_sched.h)
    // CRC-aided stop (nrldpc_cfg.early_term = 2): crc_tab[i] = x^(crc_bits-1-i) mod g for the first crc_bits information bits of a
    // codeword (payload + CRC of one code block, NRLDPCDecoder.m:298-301,336); null / 0 = parity-check stop only
    const uint32_t* crc_tab;
    int32_t batch, Z, n_layers, max_iter, ncw, sbw;
    int32_t early_term, need_ext, llr_kind;
    int32_t crc_bits;
    float alpha, scale, inv_scale;
    float beta;          // offset in grid units: message magnitude = clamp(rint(alpha*m - beta), 0, 127)
    // one int of device memory per launch in flight (the handle's ring of them): the batch counter of the parity-stop kernels that
    // refill their codeword slots (nrldpc_decode_z64p.h); null = none (every workgroup decodes its own codewords and leaves)
    int32_t* work;
    int32_t refill_mask; // those kernels take refills when (iteration & refill_mask) == refill_mask: 0 every iteration, 1 every second one
    int32_t reserved_;
};

hipError_t launch_decode(int bg, const DecArgs& a, int threads, size_t lds_bytes, hipStream_t stream);
bool has_z64_kernel(int bg, int Z); // a compile-time-Z specialisation serves this (BG, Z)
// many (Z) configurations of one base graph in one launch of the run-time-Z kernel: d_tab[nb] argument blocks,
// d_start[nb+1] first workgroup of each configuration (d_start[nb] = grid)
hipError_t launch_decode_multi(int bg, int llr_kind, const DecArgs* d_tab, const int32_t* d_start, int nb, int grid,
                               size_t lds_bytes, hipStream_t stream);
// compile-time-Z specialisations (nrldpc_decode_z64_inst.hip), one per (BG, Z): every Z that splits into waves of
// at least 40 rows where that beats the run-time-Z kernel (measured, tools/bench_all_z.py)
#define NRLDPC_Z64_LIST(X) \
    X(1, 60) X(1, 64) X(1, 104) X(1, 112) X(1, 120) X(1, 128) X(1, 144) X(1, 176) X(1, 192) X(1, 208) X(1, 224) X(1, 240) X(1, 256) X(1, 288) X(1, 320) X(1, 352) X(1, 384) \
    X(2, 52) X(2, 60) X(2, 64) X(2, 88) X(2, 96) X(2, 104) X(2, 112) X(2, 120) X(2, 128) X(2, 144) X(2, 192) X(2, 208) X(2, 224) X(2, 240) X(2, 256) X(2, 288) X(2, 320) X(2, 352) X(2, 384)
#define NRLDPC_Z64_DECL(bg, z) hipError_t launch_decode_z64_##bg##_##z(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64_LIST(NRLDPC_Z64_DECL)
#undef NRLDPC_Z64_DECL
// packed-geometry compile-time-Z builds (nrldpc_decode_z64p.h, nrldpc_decode_z64p_inst.hip): several codewords per wave, the
// lifting sizes <= 32; every row active and hard output only -- other calls stay with the run-time-Z kernel
#define NRLDPC_Z64P_LIST(X) \
    X(1, 2) X(1, 3) X(1, 4) X(1, 5) X(1, 6) X(1, 7) X(1, 8) X(1, 9) X(1, 10) X(1, 11) X(1, 12) X(1, 13) X(1, 14) X(1, 15) X(1, 16) X(1, 18) X(1, 20) X(1, 22) X(1, 24) X(1, 26) X(1, 28) X(1, 30) X(1, 32) X(1, 36) X(1, 40) X(1, 44) X(1, 48) X(1, 52) X(1, 56) X(1, 72) X(1, 80) X(1, 88) X(1, 96) X(1, 176) X(1, 352) \
    X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 7) X(2, 8) X(2, 9) X(2, 10) X(2, 11) X(2, 12) X(2, 13) X(2, 14) X(2, 15) X(2, 16) X(2, 18) X(2, 20) X(2, 22) X(2, 24) X(2, 26) X(2, 28) X(2, 30) X(2, 32) X(2, 36) X(2, 40) X(2, 44) X(2, 48) X(2, 52) X(2, 56) X(2, 72) X(2, 80)
// ... of which these serve fixed iteration counts only: with the parity-check stop the block-geometry split build of the
// same size is faster (BG2 Z = 52: 0.44 against 0.54 ms; at 25 fixed iterations the packed build wins by 9 %)
// (BG1 Z = 176: 1.585 against 1.485 ms with the parity stop, 3.98 against 4.17 ms at 25 fixed iterations)
#define NRLDPC_Z64P_NOT_ET(X) X(2, 52) X(1, 176)
#define NRLDPC_Z64P_DECL(bg, z) hipError_t launch_decode_z64p_##bg##_##z(const DecArgs& a, hipStream_t stream); \
    hipError_t launch_decode_z64pg_##bg##_##z(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64P_LIST(NRLDPC_Z64P_DECL)
#undef NRLDPC_Z64P_DECL
bool has_z64p_kernel(int bg, int Z, bool early_term);
// ... and the INTERLEAVED block geometry (NRLDPC_Z64I_LIST): nrldpc_dispatch_lists.h -- a header of its own, read only by the dispatch
// (nrldpc_decode.hip), so that a change of an entry's mode bits recompiles that entry's unit and the dispatch, not every kernel
// ... and the packed geometry's pipelined one-thread-per-row builds (nrldpc_decode_z64p.h, MODE 1 / 2; -DNRLDPC_Z64P_ROW): (BG, Z,
// row waves per workgroup).  EMPTY: built for BG2's large lifting sizes that do not split into full waves (88, 96, 176, 352 with
// 6 row waves; 144, 160, 288, 320 with 5), bit-exact, and slower than the block-geometry kernels at every one of them
// (profiles/r04_packed_row_bg2.txt: +5...+25 % at 25 fixed iterations, +3...+56 % with the parity stop) -- the doubled rings
// leave a CU three 6-wave workgroups where Z = 384's block geometry holds four.  The kernel modes stay for the next shape.
#define NRLDPC_Z64PR_LIST(X)
#define NRLDPC_Z64PR_DECL(bg, z, rw) hipError_t launch_decode_z64pr_##bg##_##z(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64PR_LIST(NRLDPC_Z64PR_DECL)
#undef NRLDPC_Z64PR_DECL
// ... and pruned layer counts with packed builds of their own: BASELINE.json configs[0] (BG2, A = 100, R = 1/3: Z = 20, 12 rows),
// the operating point of the reference's plot_BLER_vs_SNR.m defaults
#define NRLDPC_Z64P_NL_LIST(X) X(2, 20, 12)
#define NRLDPC_Z64P_NL_DECL(bg, z, nl) hipError_t launch_decode_z64p_##bg##_##z##_nl##nl(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64P_NL_LIST(NRLDPC_Z64P_NL_DECL)
#undef NRLDPC_Z64P_NL_DECL
// pruned layer counts with software-pipelined builds of their own (one translation unit each): the active-layer counts
// of BASELINE.json's rate-matching sweep at BG2 Z=384 (R = 1/4 ... 2/3 -> 32, 22, 17, 12, 9, 7 rows; R = 1/5 is all 42),
// of its BG1 Z=384 R=8/9 shard (5 rows) and of BG1 R = 2/3 and 1/2 (13, 24); BG2 Z=208 with 21 rows is the operating point of the
// reference's plot_BLER_vs_SNR.m defaults (A = 3842, R = 1/3: two code blocks).  Every other count runs the general kernel.
#define NRLDPC_Z64_NL_LIST(X) \
    X(1, 384, 5) X(1, 384, 13) X(1, 384, 24) X(2, 384, 32) X(2, 384, 22) X(2, 384, 17) X(2, 384, 12) X(2, 384, 9) X(2, 384, 7) X(2, 208, 21)
#define NRLDPC_Z64_NL_DECL(bg, z, nl) hipError_t launch_decode_z64_##bg##_##z##_nl##nl(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64_NL_LIST(NRLDPC_Z64_NL_DECL)
#undef NRLDPC_Z64_NL_DECL

struct EncArgs {
    const uint8_t* info; // [batch][kb*Z]
    uint8_t* cw;         // [batch][ncols*Z]
    const uint16_t* row_ptr; // device copies of the base graph (CSR) with shifts reduced mod Z
    const uint8_t* col;
    const uint16_t* shift;
    int32_t batch, Z, nrows, ncols, kb, nnz;
    int32_t p0_shift;        // rot(p0, p0_shift) = lam0+lam1+lam2+lam3
    int32_t step_row[3];     // substitution order for the other three core-parity blocks
    int32_t step_col[3];     // unknown block solved at each step (0..3 relative to kb)
    int32_t step_shift[3];   // shift of the unknown block in that row
    int32_t step_nk[3];      // core-parity blocks of that row already known at that step ...
    int32_t step_kcol[3][3]; // ... their columns (0..3 relative to kb) ...
    int32_t step_kshift[3][3]; // ... and shifts
};

hipError_t launch_encode(const EncArgs& a, hipStream_t stream);

#define NRLDPC_MAX_C 160

struct RmArgs {
    const float* g;   // [n_tb][G] demodulator LLRs
    float* harq;      // [n_tb][C][N_cb] soft buffer, accumulated in place; null when I_HARQ == 0
    void* out;        // [n_tb*C][2Z+N] f32 or f16
    int32_t out_f16;
    int32_t n_tb, C, G, Z, K, Kp, N, N_cb, k0, Qm;
    int32_t E[NRLDPC_MAX_C];   // E_r
    int32_t off[NRLDPC_MAX_C]; // offset of code block r inside g_tilde
};
hipError_t launch_rate_recover(const RmArgs& a, hipStream_t stream);

struct CrcPlan {
    uint32_t poly;            // generator with the x^L term
    int32_t L, chunk;         // CRC length, bits per lane (64*chunk >= message length)
    uint32_t shiftmat[6][24]; // level s: column b = x^(b + chunk*2^s) mod g
    uint32_t horner[24];      // column b = x^(b + segment length) mod g: folds per-code-block remainders
    uint32_t horner_tail[24]; // same for the last (shorter) segment of the transmit side
};
struct CrcArgs {
    const uint8_t* c_hat;     // [n_tb*C][K] decoded code blocks
    uint8_t* b_hat;           // [n_tb][B]: payload + transport-block CRC; a_hat = first A bytes of a row
    int32_t* ok;              // [n_tb]
    int32_t* cb_pass;         // [n_tb][C] or null
    int32_t n_tb, C, K, Kp, Lcb, A, B;
    // reference state machine (NRLDPCDecoder.m:286-314,337): a code block's payload is written (and its pass flag
    // set) only when its CRC holds AND its CBGTI flag is 1; otherwise b_hat keeps its content (keep_b_hat, I_HARQ != 0)
    // or is zero (:290); pass flags are sticky across calls when `sticky` (they are in/out then)
    int32_t keep_b_hat, sticky;
    uint8_t cbgti[NRLDPC_MAX_C]; // CBGTI_flags (NRLDPC.m:471-477), 1 = (re)transmitted
    CrcPlan cb, tb;
};
hipError_t launch_crc_check(const CrcArgs& a, hipStream_t stream);

struct CrcAttachArgs {
    const uint8_t* a;   // [n_tb][A] payload bits
    uint8_t* c;         // [n_tb*C][K] code blocks: payload (+ TB CRC) | CB CRC | zero fillers
    int32_t n_tb, C, K, Kp, Lcb, A, B;
    CrcPlan cb, tb;     // both sized for one code block's payload K'-L_cb (the TB CRC is folded from segments)
};
hipError_t launch_crc_attach(const CrcAttachArgs& a, hipStream_t stream);

struct TxRmArgs {
    const uint8_t* cw;  // [n_tb*C][2Z+N] encoded code blocks
    uint8_t* g;         // [n_tb][G] rate-matched, interleaved, concatenated bits
    int32_t n_tb, C, G, Z, K, Kp, N, N_cb, k0, Qm;
    int32_t E[NRLDPC_MAX_C];
    int32_t off[NRLDPC_MAX_C];
};
hipError_t launch_rate_match(const TxRmArgs& a, hipStream_t stream);

struct ChanArgs {
    const uint8_t* g;   // [n_sym * Qm] bits, one per byte
    float* llr;         // [n_sym * Qm] exact LLRs, positive = bit 0
    int64_t n_sym;
    uint64_t seed, first_symbol;
    int32_t Qm;
    float sigma, inv_n0, inv_norm; // sqrt(N0/2); 1/N0; 1/sqrt(2 mean(level^2)) of one rail
};
hipError_t launch_awgn_llr(const ChanArgs& a, hipStream_t stream);

// int8 wire format of the host path (nrldpc_host_quant.h) -> fp16 LLRs the decoder kernels ingest (nrldpc_expand.hip)
hipError_t launch_expand_i8(const int8_t* d_q, void* d_out_f16, size_t n, float inv_scale, hipStream_t stream);

} // namespace nrldpc
#endif
