/*
 * nrldpc_chain_oracle.c -- CPU ORACLE (test infrastructure) for the stages either side of the LDPC core:
 * literal restatements of the reference's per-element loops, used to check the device kernels of rows
 * N1 (rate recovery) and N2 (CRC) of SURVEY.md section 8f.
 *
 *   orc_rate_recover : NRLDPCDecoder.m:143-169 (code_block_concatenation), :172-197 (bit_interleaving),
 *                      :200-242 (bit_selection incl. HARQ soft buffer) and :262-264 (2Z zero prefix, NaN -> +inf),
 *                      in fp32 with the reference's accumulation order (k ascending).
 *   orc_rate_match   : NRLDPCEncoder.m:168-196 (bit_selection, NaN fillers skipped), :199-226 (bit_interleaving) and
 *                      :229-256 (code_block_concatenation): the transmit-side tail, for the randomized testbench.m sweep.
 *   orc_crc          : comm.CRCGenerator / comm.CRCDetector semantics (zero initial state, no reflection, no
 *                      final XOR) for the polynomials of get_3gpp_crc_polynomial.m:3-14, bit-serial.
 * PARITY STATUS: these stages are pinned against the reference's own loops line by line; the only numeric
 * difference is fp32 instead of double accumulation of repeated LLRs (stated in DESIGN.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* g_tilde: [n_tb][G] float.  harq: [n_tb][C][N_cb] float or NULL (I_HARQ = 0).
 * out: [n_tb*C][2Z + N] float: what the decoder core receives (0 prefix, +inf fillers). */
int orc_rate_recover(int Z, int C, int K, int K_prime, int N, int N_cb, int k_0, int Q_m, int G,
                     const int32_t* E_r, const float* g_tilde, int n_tb, float* harq, float* out) {
    const int ncwz = 2 * Z + N;
    float* f = (float*)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1));
    float* e = (float*)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1));
    float* d = (float*)malloc(sizeof(float) * (size_t)N);
    uint8_t* nan_ = (uint8_t*)malloc((size_t)N);
    for (int tb = 0; tb < n_tb; ++tb) {
        int k = 0; /* NRLDPCDecoder.m:157-168: walk g_tilde once, block after block */
        for (int r = 0; r < C; ++r) {
            const int E = E_r[r];
            for (int j = 0; j < E; ++j) f[j] = g_tilde[(size_t)tb * G + k++];
            /* :191-195  e(i*E/Qm + j) = f(i + j*Qm) */
            for (int j = 0; j < E / Q_m; ++j)
                for (int i = 0; i < Q_m; ++i) e[i * (E / Q_m) + j] = f[i + j * Q_m];
            /* :223-234 */
            for (int p = 0; p < N; ++p) { d[p] = 0.0f; nan_[p] = 0; }
            int lo = K_prime - 2 * Z; if (lo < 0) lo = 0;
            for (int p = lo; p < K - 2 * Z; ++p) nan_[p] = 1;
            int kk = 0, j = 0;
            while (kk < E) {
                const int pos = (k_0 + j) % N_cb;
                if (!nan_[pos]) { d[pos] = d[pos] + e[kk]; ++kk; }
                ++j;
            }
            /* :236-239 */
            if (harq) {
                float* hb = harq + ((size_t)tb * C + r) * N_cb;
                for (int p = 0; p < N_cb; ++p) { d[p] = d[p] + hb[p]; hb[p] = d[p]; }
            }
            /* :262-264 */
            float* o = out + ((size_t)tb * C + r) * ncwz;
            for (int p = 0; p < 2 * Z; ++p) o[p] = 0.0f;
            for (int p = 0; p < N; ++p) o[2 * Z + p] = nan_[p] ? INFINITY : d[p];
        }
    }
    free(f); free(e); free(d); free(nan_);
    return 0;
}

/* cw: [n_tb*C][2Z + N] encoded code blocks (bytes 0/1); positions [K'-2Z, K-2Z) of d = cw[2Z:] are fillers (NaN in the
 * reference, NRLDPCEncoder.m:160).  g: [n_tb][G]. */
int orc_rate_match(int Z, int C, int K, int K_prime, int N, int N_cb, int k_0, int Q_m, int G, const int32_t* E_r,
                   const uint8_t* cw, int n_tb, uint8_t* g) {
    const int ncwz = 2 * Z + N;
    uint8_t* e = (uint8_t*)malloc((size_t)(G > 0 ? G : 1));
    uint8_t* f = (uint8_t*)malloc((size_t)(G > 0 ? G : 1));
    int lo = K_prime - 2 * Z; if (lo < 0) lo = 0;
    for (int tb = 0; tb < n_tb; ++tb) {
        int kg = 0;
        for (int r = 0; r < C; ++r) {
            const uint8_t* d = cw + ((size_t)tb * C + r) * ncwz + 2 * Z;
            const int E = E_r[r];
            int k = 0, j = 0; /* :186-195 */
            while (k < E) {
                const int pos = (k_0 + j) % N_cb;
                if (!(pos >= lo && pos < K - 2 * Z)) { e[k] = d[pos]; ++k; }
                ++j;
            }
            for (int jj = 0; jj < E / Q_m; ++jj) /* :219-223 */
                for (int i = 0; i < Q_m; ++i) f[i + jj * Q_m] = e[i * (E / Q_m) + jj];
            for (int q = 0; q < E; ++q) g[(size_t)tb * G + kg++] = f[q]; /* :243-253 */
        }
    }
    free(e); free(f);
    return 0;
}

/* CRC remainder (L <= 24 bits, returned MSB-first in the low bits) of `len` bits (one per byte). */
uint32_t orc_crc(uint32_t poly, int L, const uint8_t* bits, int len) {
    uint32_t reg = 0;
    const uint32_t top = 1u << (L - 1), mask = (L == 32) ? 0xFFFFFFFFu : ((1u << L) - 1);
    for (int i = 0; i < len; ++i) {
        const uint32_t fb = ((reg & top) ? 1u : 0u) ^ (bits[i] & 1u);
        reg = (reg << 1) & mask;
        if (fb) reg ^= (poly & mask);
    }
    return reg;
}
