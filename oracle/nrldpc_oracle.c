/*
 * nrldpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of robmaunder/ldpc-3gpp-matlab: the LDPC decoder core
 * reached at NRLDPCDecoder.m:265 (constructed at :120), the code construction it is parameterised
 * by (get_3gpp_base_graph.m, get_pcm.m, get_3gpp_lifting_size.m, get_3gpp_set_index.m), the
 * parameter chain of NRLDPC.m:297-543, and the stages either side of the core
 * (NRLDPCDecoder.m:143-242, 271-340; NRLDPCEncoder.m:70-256).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  Nothing under ldpc-3gpp-matlab_amd/ links or imports it.
 *
 * PARITY STATUS: "parity unpinned" for the decoder arithmetic.  The reference delegates the
 * message passing to MathWorks' closed-source comm.LDPCDecoder (NRLDPCDecoder.m:120), ships no
 * decoder test, no golden vectors and no result files (results/ is empty), and neither MATLAB nor
 * Octave exists in the build image.  What IS pinned here:
 *   - base-graph tables: structural invariants of TS 38.212 Tables 5.3.2-2/-3 (dims, nnz, row
 *     degrees, per-set maximum shift) and an entry-by-entry comparison with the reference's
 *     get_3gpp_base_graph.m when /root/reference is present (tests/test_tables.py);
 *   - encoder: H*c = 0 for every (BG, Z) -- the systematic codeword is unique, so any encoder that
 *     satisfies it is bit-identical to comm.LDPCEncoder (NRLDPCEncoder.m:158);
 *   - parameter chain: the known answers listed in SURVEY.md section 8.
 * Two decoders live here:
 *   orc_decode_nmsq      the BUILD's algorithm (layered normalised min-sum on a fixed-point grid),
 *                        in integer arithmetic; the HIP kernel must match it bit for bit.
 *   orc_decode_bp_flood  the REFERENCE's semantics (flooding sum-product in double, stop when all
 *                        parity checks hold, NRLDPCDecoder.m:38-41,120) -- the stand-in for
 *                        comm.LDPCDecoder used for BLER comparison and as the CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/nr_bg_tables.h"

#define ORC_MAX_DEG 19
#define ORC_FILL 1048576 /* 2^20: fixed-point value given to +/-inf (filler) LLRs */
#define ORC_QMAX 127

/* ------------------------------------------------------------------------------------------ */
/* Code construction                                                                          */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int bg, Z, ils;
    int nrows, ncols, kb; /* base rows, base cols, systematic base cols (22 / 10) */
    int nnz;
    const uint16_t* row_ptr;
    const uint8_t* col;
    int shift[NR_BG1_NNZ]; /* table shift mod Z  (get_pcm.m:8) */
} orc_graph;

/* get_3gpp_set_index.m:5-11 : index of the lifting-size set containing Z, or -1. */
int orc_set_index(int Z) {
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9 && nr_lifting_sets[s][k]; ++k)
            if (nr_lifting_sets[s][k] == Z) return s;
    return -1;
}

/* get_3gpp_lifting_size.m:5-16 : smallest valid Z with K_b*Z >= K', or -1. */
int orc_lifting_size(int K_b, int K_prime) {
    int best = -1;
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9 && nr_lifting_sets[s][k]; ++k) {
            int z = nr_lifting_sets[s][k];
            if (K_b * z >= K_prime && (best < 0 || z < best)) best = z;
        }
    return best;
}

/* get_3gpp_base_graph.m:1-534 + get_pcm.m:1-11 without materialising H. */
static int graph_init(orc_graph* g, int bg, int Z) {
    if (bg != 1 && bg != 2) return -1;
    int ils = orc_set_index(Z);
    if (ils < 0) return -2;
    g->bg = bg; g->Z = Z; g->ils = ils;
    if (bg == 1) {
        g->nrows = NR_BG1_ROWS; g->ncols = NR_BG1_COLS; g->kb = 22; g->nnz = NR_BG1_NNZ;
        g->row_ptr = nr_bg1_row_ptr; g->col = nr_bg1_col;
        for (int e = 0; e < g->nnz; ++e) g->shift[e] = nr_bg1_shift[ils][e] % Z;
    } else {
        g->nrows = NR_BG2_ROWS; g->ncols = NR_BG2_COLS; g->kb = 10; g->nnz = NR_BG2_NNZ;
        g->row_ptr = nr_bg2_row_ptr; g->col = nr_bg2_col;
        for (int e = 0; e < g->nnz; ++e) g->shift[e] = nr_bg2_shift[ils][e] % Z;
    }
    return 0;
}

/* Export the lifted graph as (row,col,shift) triples for tests. Returns nnz. */
int orc_graph_edges(int bg, int Z, int* rows, int* cols, int* shifts) {
    orc_graph g;
    if (graph_init(&g, bg, Z)) return -1;
    for (int r = 0; r < g.nrows; ++r)
        for (int e = g.row_ptr[r]; e < g.row_ptr[r + 1]; ++e) {
            rows[e] = r; cols[e] = g.col[e]; shifts[e] = g.shift[e];
        }
    return g.nnz;
}

/* Number of unsatisfied parity checks among the first n_layers base rows (0 = all).
 * Check (i,r) involves variable (j,(r+P_ij) mod Z)  (get_pcm.m:8). */
int orc_syndrome_weight(int bg, int Z, int n_layers, const uint8_t* cw) {
    orc_graph g;
    if (graph_init(&g, bg, Z)) return -1;
    if (n_layers <= 0 || n_layers > g.nrows) n_layers = g.nrows;
    int bad = 0;
    for (int i = 0; i < n_layers; ++i)
        for (int r = 0; r < Z; ++r) {
            int p = 0;
            for (int e = g.row_ptr[i]; e < g.row_ptr[i + 1]; ++e)
                p ^= cw[g.col[e] * Z + (r + g.shift[e]) % Z] & 1;
            bad += p;
        }
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* Encoder core (stands in for comm.LDPCEncoder, NRLDPCEncoder.m:49,158)                       */
/* info: K = kb*Z bits (0/1).  cw: ncols*Z bits, systematic [info; parity].                    */
/* ------------------------------------------------------------------------------------------ */
static void encode_one(const orc_graph* g, const uint8_t* info, uint8_t* cw) {
    const int Z = g->Z, kb = g->kb;
    memcpy(cw, info, (size_t)kb * Z);
    memset(cw + (size_t)kb * Z, 0, (size_t)(g->ncols - kb) * Z);
    /* lam[i] = sum over systematic columns of row i (i = 0..3) */
    uint8_t* lam = (uint8_t*)calloc((size_t)4 * Z, 1);
    for (int i = 0; i < 4; ++i)
        for (int e = g->row_ptr[i]; e < g->row_ptr[i + 1]; ++e)
            if (g->col[e] < kb)
                for (int r = 0; r < Z; ++r) lam[i * Z + r] ^= cw[g->col[e] * Z + (r + g->shift[e]) % Z];
    /* first core-parity column kb: three entries in rows 0..3; two of the shifts cancel when the
     * four rows are summed, leaving rot(p0, a) = lam0+lam1+lam2+lam3. */
    int cnt[4] = {0, 0, 0, 0}, sh0[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int e = g->row_ptr[i]; e < g->row_ptr[i + 1]; ++e)
            if (g->col[e] == kb) { cnt[i] = 1; sh0[i] = g->shift[e]; }
    int a = -1;
    for (int i = 0; i < 4; ++i) {
        if (!cnt[i]) continue;
        int n = 0;
        for (int k = 0; k < 4; ++k) n += (cnt[k] && sh0[k] == sh0[i]);
        if (n & 1) a = sh0[i];
    }
    uint8_t* p0 = cw + (size_t)kb * Z;
    for (int r = 0; r < Z; ++r) {
        uint8_t s = lam[r] ^ lam[Z + r] ^ lam[2 * Z + r] ^ lam[3 * Z + r];
        p0[(r + a) % Z] = s; /* s_r = p0[(r+a) mod Z] */
    }
    /* remaining three core-parity blocks by substitution: pick a row with one unknown block */
    int known[4] = {1, 0, 0, 0};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = 0; i < 4; ++i) {
            int unk = -1, nunk = 0, ush = 0;
            for (int e = g->row_ptr[i]; e < g->row_ptr[i + 1]; ++e) {
                int c = g->col[e] - kb;
                if (c >= 0 && c < 4 && !known[c]) { unk = c; ush = g->shift[e]; ++nunk; }
            }
            if (nunk != 1) continue;
            uint8_t* pu = cw + (size_t)(kb + unk) * Z;
            for (int r = 0; r < Z; ++r) {
                uint8_t s = lam[i * Z + r];
                for (int e = g->row_ptr[i]; e < g->row_ptr[i + 1]; ++e) {
                    int c = g->col[e] - kb;
                    if (c >= 0 && c < 4 && known[c]) s ^= cw[g->col[e] * Z + (r + g->shift[e]) % Z];
                }
                pu[(r + ush) % Z] = s;
            }
            known[unk] = 1;
            break;
        }
    }
    free(lam);
    /* extension rows: the row's own parity column (identity, shift 0) equals the row sum */
    for (int i = 4; i < g->nrows; ++i) {
        uint8_t* pe = cw + (size_t)(kb + i) * Z;
        for (int r = 0; r < Z; ++r) {
            uint8_t s = 0;
            for (int e = g->row_ptr[i]; e < g->row_ptr[i + 1]; ++e)
                if (g->col[e] < kb + 4) s ^= cw[g->col[e] * Z + (r + g->shift[e]) % Z];
            pe[r] = s;
        }
    }
}

int orc_encode(int bg, int Z, const uint8_t* info, int batch, uint8_t* cw) {
    orc_graph g;
    int rc = graph_init(&g, bg, Z);
    if (rc) return rc;
    const size_t K = (size_t)g.kb * Z, N = (size_t)g.ncols * Z;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < batch; ++b) encode_one(&g, info + b * K, cw + b * N);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* BUILD algorithm: layered normalised min-sum on a fixed-point grid ("NMS-Q").               */
/* All quantities are integers in units of 1/scale LLR.                                        */
/*   ingest   q_v = NaN ? 0 : rint(clamp(llr_v*scale, -127, 127)); +/-inf in a core column     */
/*            (j < kb+4) -> +/-2^20 (filler "certain" bits, NRLDPCDecoder.m:264)               */
/*   layer l, row z, edges in ascending column order, v_j = col_j*Z + (z+P_lj) mod Z :         */
/*            t_j = APP[v_j] - r[l,j,z];  m1<=m2 two smallest |t_j|;  S = xor of (t_j<0)       */
/*            M1 = clamp(rint(alpha*m1 - beta), 0, 127), M2 likewise with m2 (exact alpha*m - beta,   */
/*            rounded once to the nearest integer, ties to even; beta = 0 is plain normalised     */
/*            min-sum; beta is a multiple of half a grid unit)                                    */
/*            r'_j = ((t_j<0)^S ? -1 : +1) * (|t_j|==m1 ? M2 : M1);  APP[v_j] = t_j + r'_j     */
/*   stop     after an iteration if early_term and every parity of the active rows holds        */
/*   output   hard_k = APP_k < 0 (k < K); app = APP/scale                                      */
/* ------------------------------------------------------------------------------------------ */
/* qmax: the saturation of channel values and messages in grid units -- ORC_QMAX (127: the int8 grid of the kernels) for the
 * build algorithm; the "wide" variant (orc_decode_onmsq_wide) takes 32767 to show what the 8-bit grid costs in BLER. */
static int32_t ingest(double llr, int scale, int core, int qmax) {
    if (llr != llr) return 0;
    if (isinf(llr)) return core ? (llr > 0 ? ORC_FILL : -ORC_FILL) : (llr > 0 ? qmax : -qmax);
    float x = (float)llr * (float)scale;
    if (x > (float)qmax) x = (float)qmax;
    if (x < -(float)qmax) x = -(float)qmax;
    return (int32_t)nearbyintf(x);
}

/* alpha*m - beta is formed exactly (double holds the 24-bit alpha times the <= 21-bit m) and rounded ONCE, to the
 * nearest integer, ties to even; the kernels get the same value from one fp32 fused multiply-add against 2^23 - beta. */
static int32_t scale_mag(float alpha, float beta, int32_t m, int qmax) {
    double f = nearbyint((double)alpha * (double)m - (double)beta);
    if (f > (double)qmax) f = (double)qmax;
    if (f < 0.0) f = 0.0;
    return (int32_t)f;
}

/* CRC-aided stop (nrldpc_cfg.early_term = 2; the reference checks these CRCs after decoding, NRLDPCDecoder.m:298-301,336):
 * remainder of the first `bits` hard decisions (first bit = highest power) under the generator `poly` (with its x^L term),
 * bit-serial as get_3gpp_crc_polynomial.m's polynomials are used; the block passes when the remainder is 0 and a bit is set. */
typedef struct { uint32_t poly; int L, bits; } orc_crc_stop;
static int crc_block_ok(const int32_t* APP, const orc_crc_stop* c) {
    uint32_t r = 0, any = 0;
    for (int i = 0; i < c->bits; ++i) {
        const uint32_t b = APP[i] < 0;
        any |= b;
        r = (r << 1) | b;
        if (r >> c->L) r ^= c->poly;
    }
    return r == 0 && any;
}

static int nmsq_one(const orc_graph* g, int n_layers, int max_iter, int early_term, float alpha, float beta, int qmax,
                    const int32_t* q, uint8_t* hard, int32_t* app_q, int16_t* rmsg, int32_t* APP, const orc_crc_stop* crc) {
    const int Z = g->Z;
    const int N = g->ncols * Z;
    memcpy(APP, q, sizeof(int32_t) * (size_t)N);
    memset(rmsg, 0, sizeof(int16_t) * (size_t)g->row_ptr[n_layers] * Z);
    int it = 0;
    for (it = 1; it <= max_iter; ++it) {
        for (int l = 0; l < n_layers; ++l) {
            const int e0 = g->row_ptr[l], deg = g->row_ptr[l + 1] - e0;
            for (int z = 0; z < Z; ++z) {
                int32_t t[ORC_MAX_DEG];
                int vi[ORC_MAX_DEG];
                int32_t m1 = INT32_MAX, m2 = INT32_MAX;
                int S = 0;
                for (int j = 0; j < deg; ++j) {
                    int v = g->col[e0 + j] * Z + (z + g->shift[e0 + j]) % Z;
                    vi[j] = v;
                    t[j] = APP[v] - rmsg[(size_t)(e0 + j) * Z + z];
                    int32_t a = t[j] < 0 ? -t[j] : t[j];
                    if (a < m1) { m2 = m1; m1 = a; } else if (a < m2) m2 = a;
                    S ^= (t[j] < 0);
                }
                const int32_t M1 = scale_mag(alpha, beta, m1, qmax), M2 = scale_mag(alpha, beta, m2, qmax);
                for (int j = 0; j < deg; ++j) {
                    int32_t a = t[j] < 0 ? -t[j] : t[j];
                    int32_t mag = (a == m1) ? M2 : M1;
                    int32_t r = ((t[j] < 0) ^ S) ? -mag : mag;
                    APP[vi[j]] = t[j] + r;
                    rmsg[(size_t)(e0 + j) * Z + z] = (int16_t)r;
                }
            }
        }
        if (early_term) {
            int bad = 0;
            for (int l = 0; l < n_layers && !bad; ++l)
                for (int z = 0; z < Z && !bad; ++z) {
                    int p = 0;
                    for (int e = g->row_ptr[l]; e < g->row_ptr[l + 1]; ++e)
                        p ^= (APP[g->col[e] * Z + (z + g->shift[e]) % Z] < 0);
                    bad |= p;
                }
            if (!bad) break;
            if (crc && crc_block_ok(APP, crc)) break;
        }
    }
    if (it > max_iter) it = max_iter;
    for (int k = 0; k < g->kb * Z; ++k) hard[k] = APP[k] < 0;
    if (app_q) memcpy(app_q, APP, sizeof(int32_t) * (size_t)N);
    return it;
}

/* llr: [batch][ncols*Z] double.  hard: [batch][K] bytes.  iters: [batch] or NULL.
 * app: [batch][ncols*Z] float (APP/scale) or NULL. */
static int decode_onmsq_q(int bg, int Z, int n_layers, int max_iter, int early_term, float alpha, float beta, int scale, int qmax,
                          const double* llr, int batch, uint8_t* hard, int32_t* iters, float* app, const orc_crc_stop* crc) {
    orc_graph g;
    int rc = graph_init(&g, bg, Z);
    if (rc) return rc;
    if (n_layers <= 0 || n_layers > g.nrows) n_layers = g.nrows;
    if (n_layers < 4) return -3;
    const size_t N = (size_t)g.ncols * Z, K = (size_t)g.kb * Z;
#pragma omp parallel
    {
        int32_t* q = (int32_t*)malloc(sizeof(int32_t) * N);
        int32_t* APP = (int32_t*)malloc(sizeof(int32_t) * N);
        int32_t* aq = (int32_t*)malloc(sizeof(int32_t) * N);
        int16_t* rm = (int16_t*)malloc(sizeof(int16_t) * (size_t)g.nnz * Z);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; ++b) {
            for (size_t v = 0; v < N; ++v) q[v] = ingest(llr[b * N + v], scale, (int)(v / Z) < g.kb + 4, qmax);
            int it = nmsq_one(&g, n_layers, max_iter, early_term, alpha, beta, qmax, q, hard + b * K, aq, rm, APP, crc);
            if (iters) iters[b] = it;
            if (app)
                for (size_t v = 0; v < N; ++v) app[b * N + v] = (float)aq[v] / (float)scale;
        }
        free(q); free(APP); free(aq); free(rm);
    }
    return 0;
}

int orc_decode_onmsq(int bg, int Z, int n_layers, int max_iter, int early_term, float alpha, float beta, int scale,
                     const double* llr, int batch, uint8_t* hard, int32_t* iters, float* app) {
    return decode_onmsq_q(bg, Z, n_layers, max_iter, early_term, alpha, beta, scale, ORC_QMAX, llr, batch, hard, iters, app, NULL);
}

/* orc_decode_onmsq with early termination by parity check OR by the code block's CRC (nrldpc_cfg.early_term = 2) */
int orc_decode_onmsq_crc(int bg, int Z, int n_layers, int max_iter, float alpha, float beta, int scale, uint32_t crc_poly, int crc_len,
                         int crc_bits, const double* llr, int batch, uint8_t* hard, int32_t* iters, float* app) {
    if (crc_len < 6 || crc_len > 24 || (crc_poly >> crc_len) != 1u || crc_bits <= crc_len) return -5;
    const orc_crc_stop c = {crc_poly, crc_len, crc_bits};
    return decode_onmsq_q(bg, Z, n_layers, max_iter, 1, alpha, beta, scale, ORC_QMAX, llr, batch, hard, iters, app, &c);
}

/* The same algorithm on a WIDE grid: channel values and messages saturate at +/-qmax grid units (e.g. 32767: 16-bit
 * messages, no +/-15.9 LLR ingest clamp at scale 8) instead of the kernels' +/-127.  No kernel computes this; it exists to
 * put a number on what the 8-bit grid costs in dB (tests/test_bler_gap_gpu.py::test_cost_of_the_8_bit_grid). */
int orc_decode_onmsq_wide(int bg, int Z, int n_layers, int max_iter, int early_term, float alpha, float beta, int scale, int qmax,
                          const double* llr, int batch, uint8_t* hard, int32_t* iters, float* app) {
    if (qmax < 1 || qmax > 32767) return -4;
    return decode_onmsq_q(bg, Z, n_layers, max_iter, early_term, alpha, beta, scale, qmax, llr, batch, hard, iters, app, NULL);
}

/* beta = 0: plain normalised min-sum (the committed golden vectors of round 1 were made with it) */
int orc_decode_nmsq(int bg, int Z, int n_layers, int max_iter, int early_term, float alpha, int scale,
                    const double* llr, int batch, uint8_t* hard, int32_t* iters, float* app) {
    return orc_decode_onmsq(bg, Z, n_layers, max_iter, early_term, alpha, 0.0f, scale, llr, batch, hard, iters, app);
}

/* ------------------------------------------------------------------------------------------ */
/* REFERENCE semantics: flooding sum-product in double, stop when H*c = 0                      */
/* (comm.LDPCDecoder as configured at NRLDPCDecoder.m:120; +inf fillers, 0 punctured, :262-264) */
/* ------------------------------------------------------------------------------------------ */
static int bp_one(const orc_graph* g, int n_layers, int max_iter, const double* lam, uint8_t* hard,
                  double* r, double* APP, double* app_out) {
    const int Z = g->Z, N = g->ncols * Z;
    const int ne = g->row_ptr[n_layers];
    for (size_t i = 0; i < (size_t)ne * Z; ++i) r[i] = 0.0;
    int it;
    for (it = 1; it <= max_iter; ++it) {
        /* variable-node totals from the previous sweep's check messages */
        for (int v = 0; v < N; ++v) APP[v] = lam[v];
        for (int l = 0; l < n_layers; ++l)
            for (int e = g->row_ptr[l]; e < g->row_ptr[l + 1]; ++e)
                for (int z = 0; z < Z; ++z) APP[g->col[e] * Z + (z + g->shift[e]) % Z] += r[(size_t)e * Z + z];
        /* check-node update with q = APP - r_old (flooding: all q from the same snapshot) */
        for (int l = 0; l < n_layers; ++l) {
            const int e0 = g->row_ptr[l], deg = g->row_ptr[l + 1] - e0;
            for (int z = 0; z < Z; ++z) {
                double th[ORC_MAX_DEG], pre[ORC_MAX_DEG + 1], suf[ORC_MAX_DEG + 1];
                for (int j = 0; j < deg; ++j) {
                    int v = g->col[e0 + j] * Z + (z + g->shift[e0 + j]) % Z;
                    double a = APP[v], b = r[(size_t)(e0 + j) * Z + z];
                    double qv = isinf(a) ? a : a - b;
                    th[j] = tanh(0.5 * qv);
                }
                pre[0] = 1.0;
                for (int j = 0; j < deg; ++j) pre[j + 1] = pre[j] * th[j];
                suf[deg] = 1.0;
                for (int j = deg - 1; j >= 0; --j) suf[j] = suf[j + 1] * th[j];
                for (int j = 0; j < deg; ++j) {
                    double p = pre[j] * suf[j + 1];
                    const double lim = 1.0 - 1e-15;
                    if (p > lim) p = lim;
                    if (p < -lim) p = -lim;
                    r[(size_t)(e0 + j) * Z + z] = 2.0 * atanh(p);
                }
            }
        }
        /* a-posteriori totals, hard decision, syndrome */
        for (int v = 0; v < N; ++v) APP[v] = lam[v];
        for (int l = 0; l < n_layers; ++l)
            for (int e = g->row_ptr[l]; e < g->row_ptr[l + 1]; ++e)
                for (int z = 0; z < Z; ++z) APP[g->col[e] * Z + (z + g->shift[e]) % Z] += r[(size_t)e * Z + z];
        int bad = 0;
        for (int l = 0; l < n_layers && !bad; ++l)
            for (int z = 0; z < Z && !bad; ++z) {
                int p = 0;
                for (int e = g->row_ptr[l]; e < g->row_ptr[l + 1]; ++e)
                    p ^= (APP[g->col[e] * Z + (z + g->shift[e]) % Z] < 0);
                bad |= p;
            }
        if (!bad) break;
    }
    if (it > max_iter) it = max_iter;
    for (int k = 0; k < g->kb * Z; ++k) hard[k] = APP[k] < 0;
    if (app_out) memcpy(app_out, APP, sizeof(double) * (size_t)N);
    return it;
}

/* app (nullable): a-posteriori LLRs [batch][ncols*Z] after the last sweep (comm.LDPCDecoder's
 * 'DecisionMethod','Soft decision' output; the reference uses hard decisions only). */
int orc_decode_bp_flood_app(int bg, int Z, int n_layers, int max_iter, const double* llr, int batch,
                            uint8_t* hard, int32_t* iters, int nthreads, double* app) {
    orc_graph g;
    int rc = graph_init(&g, bg, Z);
    if (rc) return rc;
    if (n_layers <= 0 || n_layers > g.nrows) n_layers = g.nrows;
    const size_t N = (size_t)g.ncols * Z, K = (size_t)g.kb * Z;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double* r = (double*)malloc(sizeof(double) * (size_t)g.nnz * Z);
        double* APP = (double*)malloc(sizeof(double) * N);
        double* lam = (double*)malloc(sizeof(double) * N);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; ++b) {
            for (size_t v = 0; v < N; ++v) {
                double x = llr[b * N + v];
                lam[v] = (x != x) ? 0.0 : x;
            }
            int it = bp_one(&g, n_layers, max_iter, lam, hard + b * K, r, APP, app ? app + b * N : NULL);
            if (iters) iters[b] = it;
        }
        free(r); free(APP); free(lam);
    }
    return 0;
}

int orc_decode_bp_flood(int bg, int Z, int n_layers, int max_iter, const double* llr, int batch,
                        uint8_t* hard, int32_t* iters, int nthreads) {
    return orc_decode_bp_flood_app(bg, Z, n_layers, max_iter, llr, batch, hard, iters, nthreads, NULL);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
