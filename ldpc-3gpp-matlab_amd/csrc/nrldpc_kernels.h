// nrldpc_kernels.h -- argument blocks and launchers shared by the C ABI and the HIP kernels.
#ifndef NRLDPC_KERNELS_H
#define NRLDPC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NRLDPC_K_F32 0
#define NRLDPC_K_F16 1

namespace nrldpc {

struct DecArgs {
    const void* llr;     // [batch][ncols*Z] f32 or f16
    uint8_t* hard;       // [batch][kb*Z]
    int32_t* iters;      // [batch] or null
    float* app;          // [batch][ncols*Z] or null
    const int32_t* rot;  // per table edge: ring rotation P_e * sbw in bytes (nrldpc_sched.h)
    int32_t batch, Z, n_layers, max_iter, ncw, sbw;
    int32_t early_term, need_ext, llr_kind;
    float alpha, scale, inv_scale;
};

hipError_t launch_decode(int bg, const DecArgs& a, int threads, size_t lds_bytes, hipStream_t stream);
hipError_t launch_decode_z384(int bg, const DecArgs& a, hipStream_t stream); // compile-time-Z specialisation

struct EncArgs {
    const uint8_t* info; // [batch][kb*Z]
    uint8_t* cw;         // [batch][ncols*Z]
    const uint16_t* row_ptr; // device copies of the base graph (CSR) with shifts reduced mod Z
    const uint8_t* col;
    const uint16_t* shift;
    int32_t batch, Z, nrows, ncols, kb;
    int32_t p0_shift;        // rot(p0, p0_shift) = lam0+lam1+lam2+lam3
    int32_t step_row[3];     // substitution order for the other three core-parity blocks
    int32_t step_col[3];     // unknown block solved at each step (0..3 relative to kb)
    int32_t step_shift[3];   // shift of the unknown block in that row
};

hipError_t launch_encode(const EncArgs& a, hipStream_t stream);

} // namespace nrldpc
#endif
