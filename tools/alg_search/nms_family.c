/*
 * nms_family.c -- EXPERIMENT (not product, not oracle): a parameterised family of layered min-sum check-node
 * rules on the decoder's fixed-point grid, used to search for the rule that brings the BLER of the build's
 * decoder closest to flooding sum-product (the reference's semantics, NRLDPCDecoder.m:120) at equal
 * iteration caps.  tools/alg_search/search.py drives it; the winner is then restated in oracle/ and csrc/.
 *
 * Build:  gcc -O3 -march=native -fopenmp -fPIC -shared -o tools/alg_search/libnmsf.so tools/alg_search/nms_family.c -lm
 * Run:    python tools/alg_search/bpref.py headline 1024 ; python tools/alg_search/search.py headline 512 '[{"alpha":0.875,"beta":3}]'
 *
 *   per row:  m1 <= m2 two smallest |t_j|
 *             a1 = max(0, m1 - max(0, c0 - c1*(m2 - m1)))      two-min (box-plus) correction of the smallest
 *             M1 = clamp(rint(alpha[l]*a1) - beta, 0, msg_max)  sent to every edge but the arg-min
 *             M2 = clamp(rint(alpha[l]*m2) - beta, 0, msg_max)  sent to the arg-min edge(s)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/nr_bg_tables.h"

typedef struct {
    float alpha[46];
    float c0, c1, beta;
    int scale, msg_max, app_max; /* app_max 0 = unclamped */
    int fp8;                     /* 1: message magnitudes rounded to the OCP e4m3 grid (4 significant bits), ties to even */
} nms_params;

static float q_e4m3(float x) { /* x >= 0 integer-valued */
    if (x <= 16.f) return x;
    int e; frexpf(x, &e);              /* x = f * 2^e, f in [0.5,1) -> 4 significant bits: step 2^(e-4) */
    float step = ldexpf(1.f, e - 4);
    return nearbyintf(x / step) * step; /* nearbyint: ties to even */
}

typedef struct { int nrows, ncols, kb, nnz; const uint16_t* row_ptr; const uint8_t* col; int shift[NR_BG1_NNZ]; } graph;

static int set_index(int Z) {
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9 && nr_lifting_sets[s][k]; ++k)
            if (nr_lifting_sets[s][k] == Z) return s;
    return -1;
}
static void ginit(graph* g, int bg, int Z) {
    int ils = set_index(Z);
    if (bg == 1) { g->nrows = 46; g->ncols = 68; g->kb = 22; g->nnz = NR_BG1_NNZ; g->row_ptr = nr_bg1_row_ptr; g->col = nr_bg1_col;
        for (int e = 0; e < g->nnz; ++e) g->shift[e] = nr_bg1_shift[ils][e] % Z;
    } else { g->nrows = 42; g->ncols = 52; g->kb = 10; g->nnz = NR_BG2_NNZ; g->row_ptr = nr_bg2_row_ptr; g->col = nr_bg2_col;
        for (int e = 0; e < g->nnz; ++e) g->shift[e] = nr_bg2_shift[ils][e] % Z; }
}

static int one(const graph* g, int Z, int n_layers, int max_iter, const nms_params* P, const double* llr, uint8_t* hard,
               float* APP, float* rm) {
    const int N = g->ncols * Z;
    for (int v = 0; v < N; ++v) {
        double x = llr[v];
        float q;
        if (x != x) q = 0;
        else if (isinf(x)) q = (v / Z < g->kb + 4) ? (x > 0 ? 1048576.f : -1048576.f) : (x > 0 ? 127.f : -127.f);
        else { float y = (float)x * (float)P->scale; if (y > 127.f) y = 127.f; if (y < -127.f) y = -127.f; q = nearbyintf(y); }
        APP[v] = q;
    }
    memset(rm, 0, sizeof(float) * (size_t)g->row_ptr[n_layers] * Z);
    int it;
    for (it = 1; it <= max_iter; ++it) {
        for (int l = 0; l < n_layers; ++l) {
            const int e0 = g->row_ptr[l], deg = g->row_ptr[l + 1] - e0;
            for (int z = 0; z < Z; ++z) {
                float t[19]; int vi[19];
                float m1 = 1e9f, m2 = 1e9f; int S = 0;
                for (int j = 0; j < deg; ++j) {
                    int v = g->col[e0 + j] * Z + (z + g->shift[e0 + j]) % Z;
                    vi[j] = v; t[j] = APP[v] - rm[(size_t)(e0 + j) * Z + z];
                    float a = fabsf(t[j]);
                    if (a < m1) { m2 = m1; m1 = a; } else if (a < m2) m2 = a;
                    S ^= (t[j] < 0);
                }
                float corr = P->c0 - P->c1 * (m2 - m1); if (corr < 0) corr = 0;
                float a1 = m1 - corr; if (a1 < 0) a1 = 0;
                float M1 = nearbyintf(P->alpha[l] * a1) - P->beta, M2 = nearbyintf(P->alpha[l] * m2) - P->beta;
                if (M1 < 0) M1 = 0; if (M2 < 0) M2 = 0;
                if (M1 > P->msg_max) M1 = (float)P->msg_max; if (M2 > P->msg_max) M2 = (float)P->msg_max;
                if (P->fp8) { M1 = q_e4m3(M1); M2 = q_e4m3(M2); }
                for (int j = 0; j < deg; ++j) {
                    float mag = (fabsf(t[j]) == m1) ? M2 : M1;
                    float r = ((t[j] < 0) ^ S) ? -mag : mag;
                    float v = t[j] + r;
                    if (P->app_max && fabsf(v) < 1e5f) { if (v > P->app_max) v = (float)P->app_max; if (v < -P->app_max) v = -(float)P->app_max; }
                    APP[vi[j]] = v; rm[(size_t)(e0 + j) * Z + z] = r;
                }
            }
        }
        int bad = 0;
        for (int l = 0; l < n_layers && !bad; ++l)
            for (int z = 0; z < Z && !bad; ++z) {
                int p = 0;
                for (int e = g->row_ptr[l]; e < g->row_ptr[l + 1]; ++e) p ^= (APP[g->col[e] * Z + (z + g->shift[e]) % Z] < 0);
                bad |= p;
            }
        if (!bad) break;
    }
    if (it > max_iter) it = max_iter;
    for (int k = 0; k < g->kb * Z; ++k) hard[k] = APP[k] < 0;
    return it;
}

int nmsf_decode(int bg, int Z, int n_layers, int max_iter, const nms_params* P, const double* llr, int batch, uint8_t* hard,
                int32_t* iters) {
    graph g; ginit(&g, bg, Z);
    if (n_layers <= 0) n_layers = g.nrows;
    const size_t N = (size_t)g.ncols * Z, K = (size_t)g.kb * Z;
#pragma omp parallel
    {
        float* APP = malloc(sizeof(float) * N); float* rm = malloc(sizeof(float) * (size_t)g.nnz * Z);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < batch; ++b) iters[b] = one(&g, Z, n_layers, max_iter, P, llr + b * N, hard + b * K, APP, rm);
        free(APP); free(rm);
    }
    return 0;
}
