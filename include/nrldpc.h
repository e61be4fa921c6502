/*
 * nrldpc.h -- C ABI of the MI355X-native NR LDPC codec core (libnrldpc_hip.so).
 *
 * Drop-in boundary for the one hot path of robmaunder/ldpc-3gpp-matlab: the LDPC coding core that
 * the reference's System objects delegate to MathWorks toolbox objects.
 *
 *   nrldpc_create   replaces  comm.LDPCDecoder('ParityCheckMatrix',H,'MaximumIterationCount',it,
 *                             'IterationTerminationCondition','Parity check satisfied')
 *                             NRLDPCDecoder.m:120 (and the commented-out comm.gpu.LDPCDecoder seam at
 *                             :117-121), and comm.LDPCEncoder('ParityCheckMatrix',H) NRLDPCEncoder.m:49.
 *                             H is never materialised: (bg, Z) identify it (NRLDPC.m:433-440,
 *                             get_3gpp_base_graph.m, get_pcm.m).
 *   nrldpc_decode   replaces  step(obj.hLDPCDecoder, cw_tilde)   NRLDPCDecoder.m:265
 *   nrldpc_encode   replaces  step(obj.hLDPCEncoder, c)          NRLDPCEncoder.m:158
 *   nrldpc_destroy  replaces  release() of those toolbox objects
 *
 * Plain pointers and sizes only.  Host-pointer entry points are synchronous; *_dev entry points take
 * device pointers plus a hipStream_t (passed as void*) and are asynchronous on that stream.
 * A handle is not thread-safe; use one handle per GPU / host thread.
 *
 * Error convention (NRLDPC.m / NRLDPCDecoder.m use two MATLAB identifiers; the MEX gateway maps
 * return codes onto exactly those):
 *   NRLDPC_ERR_UNSUPPORTED -> 'ldpc_3gpp_matlab:UnsupportedParameters'  (callers catch and skip:
 *                             plot_BLER_vs_SNR.m:173, testbench.m:51)
 *   NRLDPC_ERR_ARG / _HIP  -> 'ldpc_3gpp_matlab:Error'                  (e.g. NRLDPCDecoder.m:149)
 */
#ifndef NRLDPC_H
#define NRLDPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRLDPC_OK 0
#define NRLDPC_ERR_UNSUPPORTED 1 /* invalid BG / lifting size / layer count / parameter combination */
#define NRLDPC_ERR_ARG 2         /* shape or pointer violation */
#define NRLDPC_ERR_HIP 3         /* HIP runtime failure (text via nrldpc_last_error) */
#define NRLDPC_ERR_NOMEM 4

/* LLR element types accepted at the boundary.  Codeword-contiguous layout [batch][ncols*Z], which
 * is MATLAB's column-major (ncols*Z) x batch.  Positive LLR <=> bit 0 (NRLDPCDecoder.m:262-266):
 * 0 = punctured / untransmitted, +inf = filler bit (known 0). */
#define NRLDPC_LLR_F32 0
#define NRLDPC_LLR_F16 1
#define NRLDPC_LLR_F64 2 /* host entry point only (MATLAB double); narrowed to f32 -- large batches: quantised to the
                            kernels' int8 grid, see nrldpc_quantise_llr -- on the host */

typedef struct nrldpc_codec* nrldpc_handle;

/* ABI revision of this header.  Revision 3 put `struct_size` in front of nrldpc_cfg and nrldpc_dims (revision 2 had
 * grown both at the tail -- beta; alpha, beta -- with nothing a caller built against revision 1 could be told apart by).
 * nrldpc_abi_version() returns the revision the loaded library was built with (the library is loaded by path and has
 * no SONAME; a binding checks this number at load time, as ldpc-3gpp-matlab_amd/_capi.py does).
 * Revision 4 adds nrldpc_decode_packed and the CRC-aided stop (early_term = 2; crc_poly, crc_len, crc_bits at the tail of
 * nrldpc_cfg).
 * Revision 5 makes the active layer count a property of the CALL, not of the handle: nrldpc_set_layers / nrldpc_last_layers /
 * nrldpc_count_layers / nrldpc_pool_set_layers / nrldpc_set_llr_dtype and the value NRLDPC_LAYERS_AUTO (-1) for nrldpc_cfg.n_layers -- see "Active
 * layers" below; and adds nrldpc_pool_decode_packed.  nrldpc_cfg and nrldpc_dims keep their revision-4 layout and size.
 * Revision 6 adds nrldpc_decode_packed_layers (the layer count as an ARGUMENT of one call: nothing sticks to the handle) and
 * nrldpc_pool_set_timing / nrldpc_pool_last_kernel_ms (event-pair kernel times of every shard of a pool) and
 * nrldpc_last_host_phases (the phase times of a large host-pointer call) and nrldpc_payload_bits_dev (the Monte-Carlo loop's payload
 * draw as one kernel); nothing a revision-5 caller uses changed meaning. */
#define NRLDPC_ABI_VERSION 6

/* Active layers.  The reference always decodes the full H (NRLDPCDecoder.m:120).  A base-graph row i >= 4 owns the degree-1
 * extension-parity column kb + i; when every codeword of a call holds LLR 0 in that column (not transmitted: rate matching
 * stopped below it, NRLDPCDecoder.m:216-234) the row's check-to-variable messages are identically zero and the row can be left
 * out -- exactly, for sum-product and min-sum alike -- as can every row above it that is in the same
 * state.  At the reference's own defaults (plot_BLER_vs_SNR.m:29-41: BG2, R = 1/3) that is 21 of 42 rows, at R = 8/9 on BG1 5 of 46.
 *   n_layers = 0 (NRLDPC_LAYERS_ALL): every row;  4..46 / 4..42: that many rows, the caller vouches for the zeros;
 *   NRLDPC_LAYERS_AUTO: read off the data, per call -- n = max(4, c - kb + 1) for the highest base-graph column c of the call's
 *   codewords that holds a value other than +-0 and NaN (NaN is ingested as 0).  Exact for any rv_id, repetition, LBRM or
 *   HARQ-combined buffer because nothing is assumed about how the zeros came about.  Host-pointer entry points scan the caller's
 *   array from the top column down on the copy threads (a column block that is all zero is read once and never quantised or
 *   sent); device-pointer entry points run a pre-pass kernel and read one integer back, i.e. they synchronise `stream` once
 *   before the launch.  With cfg.alpha == 0 the check-node rule follows the count in use (nrldpc_default_rule).
 *   The count under AUTO is ONE number per call -- the maximum over the call's codewords -- so with cfg.alpha == 0 the rule,
 *   and through it a codeword's hard decisions and iteration count, can depend on which other codewords share the call
 *   (leaving rows out is exact; changing the rule is not).  A caller that needs batch-independent results gives the count
 *   (nrldpc_set_layers / nrldpc_decode_packed_layers: what the patched NRLDPCDecoder.m does, from E_r, k_0 and N_cb) or a
 *   fixed rule (cfg.alpha != 0). */
#define NRLDPC_LAYERS_ALL 0
#define NRLDPC_LAYERS_AUTO (-1)

typedef struct nrldpc_cfg {
    uint32_t struct_size; /* = sizeof(nrldpc_cfg); nrldpc_create refuses any other value (NRLDPC_ERR_ARG)      */
    int32_t bg;         /* 1 or 2                                     (NRLDPC.m:28)               */
    int32_t Z;          /* lifting size Z_c, one of the 51 of Table 5.3.2-1 (NRLDPC.m:409-411)    */
    int32_t n_layers;   /* base rows to decode, 4..46 (BG1) / 4..42 (BG2); 0 = all (reference: all); NRLDPC_LAYERS_AUTO = read
                           off each call's LLRs ("Active layers" above); nrldpc_set_layers changes it between calls */
    int32_t max_iter;   /* 'MaximumIterationCount' (NRLDPCDecoder.m:41,120); 1..2000              */
    int32_t early_term; /* 0 = always max_iter iterations; 1 = stop a codeword when all active parity checks hold (the
                           reference: 'Parity check satisfied', NRLDPCDecoder.m:120); 2 = that, or when the CRC of the
                           code block holds on the hard decisions (crc_* below) -- see "CRC-aided stop"              */
    float alpha;        /* min-sum normalisation factor, 0 < alpha <= 1; 0 = choose alpha AND beta by code rate
                           (nrldpc_default_rule: the rule whose BLER sits closest to the reference's sum-product) */
    int32_t llr_scale;  /* fixed-point units per unit LLR: power of two 1..32; 0 = default 8 */
    int32_t llr_dtype;  /* NRLDPC_LLR_*                                                            */
    int32_t device_id;  /* HIP device ordinal                                                      */
    int32_t max_batch;  /* staging capacity of the host entry points; 0 = grow on demand          */
    float beta;         /* min-sum offset in LLR units (>= 0), read only when alpha != 0: message magnitude =
                           max(alpha*min - beta, 0) on the fixed-point grid; 0 = plain normalised min-sum; rounded to
                           the nearest 1/(2*llr_scale) LLR (nrldpc_get_dims reports the value in use)               */
    /* CRC-aided stop, read only when early_term == 2 (else leave 0).  The reference checks a code block's CRC AFTER decoding
     * (CRC24B per code block when C > 1, else the transport block's CRC24A / CRC16: NRLDPCDecoder.m:298-301, 336;
     * get_3gpp_crc_polynomial.m:3-14); here the same check also ends the iterations of a codeword whose information bits are
     * already right while a parity bit is not: after an iteration a codeword is done when its active parity checks hold, OR
     * when the first crc_bits hard decisions (payload followed by its CRC, K' of NRLDPC.m) leave remainder 0 under crc_poly and
     * are not all zero (an all-zero block has remainder 0 whatever was sent; with rv_id 2 / 3 the punctured systematic bits
     * start at a-posteriori 0, i.e. hard decision 0).  Hard decisions and iteration counts are those of the oracle's
     * orc_decode_onmsq_crc.  Measured against the parity-check stop: profiles/r04_crc_stop.json. */
    uint32_t crc_poly;  /* generator polynomial with its x^crc_len term, e.g. CRC24B 0x1800063, CRC24A 0x1864CFB, CRC16 0x11021 */
    int32_t crc_len;    /* L: 24 or 16 for TS 38.212 (6..24 accepted)                                   */
    int32_t crc_bits;   /* K' = payload + CRC bits of the code block, crc_len < crc_bits <= K            */
} nrldpc_cfg;

/* Dimensions implied by (bg, Z): ncols*Z LLRs in, K = kb*Z hard bits out. */
typedef struct nrldpc_dims {
    uint32_t struct_size;     /* in: = sizeof(nrldpc_dims), set by the caller before nrldpc_get_dims */
    int32_t nrows, ncols, kb; /* 46,68,22 or 42,52,10 */
    int32_t i_ls;             /* set index (get_3gpp_set_index.m) */
    int32_t K, N_cw;          /* kb*Z, ncols*Z */
    int32_t n_layers;         /* active layer count of the next call: 4..rows, or NRLDPC_LAYERS_AUTO */
    float alpha, beta;        /* resolved check-node rule (beta in LLR units); under NRLDPC_LAYERS_AUTO with cfg.alpha == 0:
                                 the rule of the last call's count (before the first call: of all rows) */
} nrldpc_dims;

int nrldpc_abi_version(void); /* NRLDPC_ABI_VERSION of the library's build */
int nrldpc_create(const nrldpc_cfg* cfg, nrldpc_handle* out);
void nrldpc_destroy(nrldpc_handle h);
int nrldpc_get_dims(nrldpc_handle h, nrldpc_dims* out);

/* Active layer count of the calls that follow (every decode entry point, nrldpc_decode_multi_dev included): NRLDPC_LAYERS_ALL,
 * 4..rows of the base graph, or NRLDPC_LAYERS_AUTO.  No device work, no reallocation: the kernels take the count as a launch
 * argument.  With cfg.alpha == 0 the check-node rule becomes nrldpc_default_rule(bg, count) -- a call gives bit for bit what a
 * handle created with that n_layers gives.  NRLDPC_ERR_UNSUPPORTED for any other value.
 * (A System object's rate is tunable between step() calls -- G, rv_id: NRLDPC.m:51-85 -- so the count is too.) */
int nrldpc_set_layers(nrldpc_handle h, int32_t n_layers);
/* LLR element type of the calls that follow (NRLDPC_LLR_*; NRLDPC_LLR_F64 for host pointers only): a gateway whose caller holds
 * `single` or `double` arrays hands each over as it is, through one handle (staging buffers grow on demand). */
int nrldpc_set_llr_dtype(nrldpc_handle h, int32_t llr_dtype);
/* The count the most recent decode call of this handle ran with (what NRLDPC_LAYERS_AUTO found); 0 before the first call. */
int nrldpc_last_layers(nrldpc_handle h, int32_t* n_layers);
/* What NRLDPC_LAYERS_AUTO finds for `batch` codewords at the HOST address llr ([batch][ncols*Z] of llr_dtype, NRLDPC_LLR_*):
 * the layer count, 4..rows; -1 on invalid arguments (text via nrldpc_last_error).  Host function, no device needed. */
int nrldpc_count_layers(int32_t bg, int32_t Z, const void* llr, int32_t batch, int32_t llr_dtype);

/* Decode `batch` codewords.  llr: [batch][ncols*Z] of cfg.llr_dtype.  hard: [batch][K] bytes in
 * {0,1} (the K x 1 logical of comm.LDPCDecoder).  iters_out (nullable): iterations executed per
 * codeword.  app_out (nullable): a-posteriori LLRs [batch][ncols*Z] float. */
int nrldpc_decode(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out,
                  float* app_out);
int nrldpc_decode_dev(nrldpc_handle h, const void* d_llr, int32_t batch, uint8_t* d_hard,
                      int32_t* d_iters_out, float* d_app_out, void* stream);
/* nrldpc_decode with BIT-PACKED hard decisions: hard_packed: [batch][ceil(K/8)] bytes, bit k of a codeword in byte k/8 at bit
 * k%8 (least significant first; the unused bits of a codeword's last byte are 0).  The bits are packed on the device, so an
 * eighth of the bytes crosses PCIe and the copy into the caller's array: the form a MEX gateway uses (matlab/nrldpc_mex.cpp
 * unpacks into the K x C logical array the reference's comm.LDPCDecoder returns, NRLDPCDecoder.m:265). */
int nrldpc_decode_packed(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out);
/* nrldpc_decode_packed with the active layer count of THIS call as an argument (0 all / 4..rows / NRLDPC_LAYERS_AUTO): the
 * handle's own count (cfg.n_layers, nrldpc_set_layers) is neither read nor changed, so a HARQ caller that gives the count for
 * one transmission and omits it for the next gets the handle's default back (ABI revision 6).  The reference's
 * seam is one call per step: step(obj.hLDPCDecoder, cw_tilde), NRLDPCDecoder.m:265 -- the patched LDPC_coding passes the
 * count it derives from E_r, k_0 and N_cb (NRLDPC.m:463-543) here. */
int nrldpc_decode_packed_layers(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out,
                                int32_t n_layers);

/* The quantisation nrldpc_decode applies to large host batches while it copies them into its pinned staging
 * buffers (so that 1 byte per LLR crosses PCIe instead of 4): dst[i] = NaN ? 0 : rint(clamp(float(src[i]) * llr_scale,
 * +-127)) as int8 -- the kernels' own ingest arithmetic, operation for operation -- with +inf (filler bits,
 * NRLDPCDecoder.m:264) as -128.  src: n values of llr_dtype (NRLDPC_LLR_*).  Returns 1 when a -inf was met (int8 has
 * no code for it; nrldpc_decode then sends that chunk in its own format), else 0.  Host function, no device needed. */
int nrldpc_quantise_llr(int8_t* dst, const void* src, int64_t n, int32_t llr_dtype, int32_t llr_scale);

/* Mixed batches: n configurations (handles of one device that may differ in base graph, lifting size, layer
 * count, iteration cap, ...) decoded with one launch per base graph and LLR type instead of n launches --
 * a small bucket alone is a one-workgroup kernel that leaves the GPU idle.  For i < n: batch[i] codewords at
 * d_llr[i] (dtype of h[i]) -> d_hard[i], iteration counts to d_iters[i] when d_iters and d_iters[i] are
 * non-null.  Results are those of n nrldpc_decode_dev calls.  The launch tables go through a ring of four
 * (pinned, device, event) slots owned by h[0], so calls in flight on different streams never share one; asynchronous on
 * `stream`.  (The reference decodes one code block per step(), NRLDPCDecoder.m:257-266; a receiver serving many
 * users holds exactly such a mix of (BG, Z_c).) */
int nrldpc_decode_multi_dev(int32_t n, const nrldpc_handle* h, const void* const* d_llr, const int32_t* batch,
                            uint8_t* const* d_hard, int32_t* const* d_iters, void* stream);

/* ---- one node, several GPUs: codeword batches shard with no collective --------------------
 * A pool owns one codec handle and one host thread per entry of device_ids (an ordinal may repeat: several logical
 * shards on one GPU).  nrldpc_pool_decode cuts the batch into n_devices * chunks_per_device contiguous chunks and the
 * threads pull them from a queue, so that shards which finish early under early termination (BASELINE.json
 * configs[4]) take more chunks; results land in place, identical to one nrldpc_decode call.  cfg->device_id is
 * ignored.  The reference decodes one code block at a time on one thread (NRLDPCDecoder.m:257-266). */
typedef struct nrldpc_pool* nrldpc_pool_handle;
int nrldpc_pool_create(const nrldpc_cfg* cfg, const int32_t* device_ids, int32_t n_devices, int32_t chunks_per_device,
                       nrldpc_pool_handle* out);
int nrldpc_pool_decode(nrldpc_pool_handle p, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out);
/* nrldpc_pool_decode with bit-packed hard decisions ([batch][ceil(K/8)], as nrldpc_decode_packed): what a MEX gateway calls */
int nrldpc_pool_decode_packed(nrldpc_pool_handle p, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out);
/* nrldpc_set_layers for every handle of the pool.  Under NRLDPC_LAYERS_AUTO the count is found ONCE per call over the whole
 * batch (host form: before the chunks are dealt; device form: every shard scans its slice, the maximum is taken, then every
 * shard launches), so the result does not depend on how the batch is cut. */
int nrldpc_pool_set_layers(nrldpc_pool_handle p, int32_t n_layers);
/* The same for data that is already ON the devices: shard i (entry i of device_ids) decodes batch[i] codewords from
 * d_llr[i] into d_hard[i] (and d_iters[i] when d_iters and d_iters[i] are non-null); every pointer of shard i is
 * device memory of device_ids[i], in cfg.llr_dtype (F32 / F16).  One host thread per shard launches on the shard's
 * own stream; the call returns when every shard's stream is idle.  Ordering: the shard streams are private, so each shard
 * first waits (hipDeviceSynchronize on its device) for everything the caller has ALREADY submitted to that device -- LLRs a
 * kernel of the caller's is still writing when the call is made are complete before the decoder reads them; work submitted
 * from another thread DURING the call is not ordered against it.  No host copy, no collective: this is the form that
 * scales with the number of GPUs (the host-pointer form above is bounded by the host's copy bandwidth). */
int nrldpc_pool_decode_dev(nrldpc_pool_handle p, const void* const* d_llr, const int32_t* batch, uint8_t* const* d_hard,
                           int32_t* const* d_iters);
/* Kernel timing for a pool (nrldpc_set_timing for every shard's handle): when enabled, each shard records a HIP event pair
 * around its decode kernel on the shard's own launch stream; nrldpc_pool_last_kernel_ms writes the duration of every shard's
 * last launch to ms[0 .. n_devices-1] (0 for a shard that had no work).  This is how a one-process, N-GPU caller -- bench.py
 * --in-process -- gets per-GPU kernel times without a process group. */
int nrldpc_pool_set_timing(nrldpc_pool_handle p, int32_t enabled);
int nrldpc_pool_last_kernel_ms(nrldpc_pool_handle p, float* ms);
int nrldpc_pool_size(nrldpc_pool_handle p); /* number of shards (n_devices of nrldpc_pool_create) */
/* codewords each shard (entry of device_ids) decoded in the last nrldpc_pool_decode call; counts: [n_devices] */
int nrldpc_pool_last_split(nrldpc_pool_handle p, int32_t* counts);
void nrldpc_pool_destroy(nrldpc_pool_handle p);

/* Systematic encode.  info: [batch][K] bytes {0,1}; cw: [batch][ncols*Z] bytes {0,1} = [info;parity]
 * with H*cw = 0 (NRLDPCEncoder.m:158). */
int nrldpc_encode(nrldpc_handle h, const uint8_t* info, int32_t batch, uint8_t* cw);
int nrldpc_encode_dev(nrldpc_handle h, const uint8_t* d_info, int32_t batch, uint8_t* d_cw, void* stream);

/* ---- stages either side of the core; stateless, current HIP device ---- */
#define NRLDPC_MAX_C 160 /* code blocks per transport block */

/* Parameters of one transport block as the reference derives them (NRLDPC.m:297-543). */
typedef struct nrldpc_tb_params {
    int32_t bg, Z;           /* BG, Z_c */
    int32_t A, B, C;         /* payload, payload + TB CRC, code blocks */
    int32_t K, K_prime, N;   /* per code block: K, K', N */
    int32_t N_cb, k_0;       /* circular buffer length and start (NRLDPC.m:463-469, 510-543) */
    int32_t Q_m, G;          /* bits per symbol, total rate-matched bits */
    int32_t tb_crc_len;      /* 16 or 24 (CRC16 / CRC24A, NRLDPC.m:297-303) */
    int32_t cb_crc_len;      /* 0 or 24 (CRC24B when C > 1, NRLDPC.m:347-353) */
    int32_t E_r[NRLDPC_MAX_C]; /* NRLDPC.m:485-507 */
} nrldpc_tb_params;

/* Rate recovery: replaces code_block_concatenation + bit_interleaving + bit_selection (NRLDPCDecoder.m:143-242)
 * and the 2Z-zero prefix / NaN->+inf of LDPC_coding (:262-264).  d_g_tilde: [n_tb][G] f32 LLRs.
 * d_harq (nullable = I_HARQ 0): [n_tb][C][N_cb] f32 soft buffer, accumulated in place (:236-239).
 * d_cw_llr: [n_tb*C][ncols*Z] of out_dtype (NRLDPC_LLR_F32 / _F16): the decoder core's input. */
int nrldpc_rate_recover_dev(const nrldpc_tb_params* p, const float* d_g_tilde, int32_t n_tb, float* d_harq,
                            void* d_cw_llr, int32_t out_dtype, void* stream);

/* CRC stages: replaces code_block_segmentation + crc_calculation of the decoder (NRLDPCDecoder.m:271-340).
 * d_c_hat: [n_tb*C][K] hard bits from nrldpc_decode_dev.  d_b_hat: [n_tb][B] bytes (a_hat = first A of a row).
 * d_ok: [n_tb], 0 where the reference returns [] (TB CRC or any CB CRC failed).  d_cb_pass (nullable):
 * [n_tb][C] per-code-block CRC flags (code_block_CRC_passed, :95). */
int nrldpc_crc_check_dev(const nrldpc_tb_params* p, const uint8_t* d_c_hat, int32_t n_tb, uint8_t* d_b_hat,
                         int32_t* d_ok, int32_t* d_cb_pass, void* stream);

/* The same stage with the reference's state machine (NRLDPCDecoder.m:283-316,337), for incremental redundancy
 * (I_HARQ ~= 0) and code-block-group retransmission (CBGTI):
 *   cbgti_flags (host, C bytes, nullable = all 1): CBGTI_flags of NRLDPC.m:471-477; a code block is taken over --
 *       payload written to d_b_hat, pass flag set -- only when its CRC holds (C > 1) AND its flag is 1 (:304);
 *   keep_b_hat != 0 (I_HARQ ~= 0): d_b_hat is in/out, the b_hat_buffer of :286-287,311-313 -- segments of blocks
 *       not taken over keep what an earlier step stored; 0: those segments are zeroed (:289);
 *   d_cb_pass (required): in/out, the sticky code_block_CRC_passed of :280,305,315 -- the caller zeroes it at
 *       reset() (:355); d_ok[tb] = TB CRC of the resulting b_hat holds and every flag is set (:337).
 * nrldpc_crc_check_dev is this call with no CBGTI, keep_b_hat = 0 and d_cb_pass as a plain output. */
int nrldpc_crc_check_harq_dev(const nrldpc_tb_params* p, const uint8_t* d_c_hat, int32_t n_tb, uint8_t* d_b_hat,
                              int32_t* d_ok, int32_t* d_cb_pass, const uint8_t* cbgti_flags, int32_t keep_b_hat,
                              void* stream);

/* Transmit-side counterparts (vector generation for the Monte-Carlo harness, plot_BLER_vs_SNR.m:129):
 * nrldpc_crc_attach_dev replaces crc_calculation + code_block_segmentation of the encoder
 * (NRLDPCEncoder.m:70-124): d_a [n_tb][A] bits -> d_c [n_tb*C][K] code blocks (fillers 0), ready for
 * nrldpc_encode_dev.  nrldpc_rate_match_dev replaces bit_selection + bit_interleaving +
 * code_block_concatenation (NRLDPCEncoder.m:168-256): d_cw [n_tb*C][ncols*Z] -> d_g [n_tb][G]. */
int nrldpc_crc_attach_dev(const nrldpc_tb_params* p, const uint8_t* d_a, int32_t n_tb, uint8_t* d_c, void* stream);
int nrldpc_rate_match_dev(const nrldpc_tb_params* p, const uint8_t* d_cw, int32_t n_tb, uint8_t* d_g, void* stream);

/* Channel leg of the Monte-Carlo loop, fused: replaces step(hMod,g), step(hChan,tx),
 * step(hDemod,rx) of plot_BLER_vs_SNR.m:130-132 -- NRModulator.m:73-81 (TS 38.211 maps, unit average power),
 * comm.AWGNChannel at EsN0_dB (:50,105), NRDemodulator.m:76-84 (exact LLRs, Variance = 10^(-EsN0/10), :106).
 * d_g: n_bits rate-matched bits (bytes {0,1}), n_bits a multiple of Q_m in {1,2,4,6,8}; d_g_tilde: n_bits f32 LLRs,
 * positive = bit 0.  Noise: Philox-4x32-10 keyed by `seed`, counter = first_symbol + symbol index, Box-Muller --
 * the same (seed, symbol) always sees the same noise, whatever the batch split.  Current HIP device. */
int nrldpc_awgn_llr_dev(const uint8_t* d_g, int64_t n_bits, int32_t Q_m, float EsN0_dB, uint64_t seed,
                        uint64_t first_symbol, float* d_g_tilde, void* stream);
/* Payload of the Monte-Carlo loop: replaces `a = round(rand(A,1))` of plot_BLER_vs_SNR.m:118 for n_tb transport blocks at once.
 * d_a: [n_tb][A] bytes {0,1}; bit i of the block with GLOBAL index first_block + b is bit (i mod 64) of
 * splitmix64(seed + ((first_block + b) * ceil(A/64) + i div 64 + 1) * 0x9E3779B97F4A7C15) -- the same (seed, block) always draws the
 * same payload, whatever the batch split (MATLAB's Mersenne-Twister stream cannot be replicated outside MATLAB: SURVEY.md 8c; curves
 * are compared statistically).  Current HIP device.  (ABI revision 6.) */
int nrldpc_payload_bits_dev(uint64_t seed, uint64_t first_block, int32_t n_tb, int32_t A, uint8_t* d_a, void* stream);

/* Kernel timing: when enabled, every *_dev / host call records HIP events around its kernel on the
 * launch stream; nrldpc_last_kernel_ms synchronises on the stop event and returns the duration. */
int nrldpc_set_timing(nrldpc_handle h, int32_t enabled);
int nrldpc_last_kernel_ms(nrldpc_handle h, float* ms);
/* Where the handle's last LARGE host-pointer call (nrldpc_decode / nrldpc_decode_packed[_layers] above 8 MB: the chunked pipeline)
 * spent its time -- what NRLDPC_HOST_TRACE=1 prints, as numbers (ABI revision 6): out10[0] chunks, [1] codewords per chunk, [2] layers
 * decoded, then milliseconds of the CALLER's thread: [3] NRLDPC_LAYERS_AUTO scan, [4] copy / quantise into pinned memory incl. the
 * H2D enqueue, [5] kernel launch + D2H enqueue, [6] waiting for the device, [7] copying results out; [8] NUMA node the copy
 * threads are pinned to, [9] CPU the caller ran on.  NRLDPC_ERR_ARG when no such call has been made on the handle. */
int nrldpc_last_host_phases(nrldpc_handle h, double* out10);

/* Check-node rule nrldpc_create applies when cfg.alpha == 0, by base graph and active layer count (0 = all):
 * the (alpha, beta) pair measured closest to flooding sum-product (the reference's comm.LDPCDecoder,
 * NRLDPCDecoder.m:120) at equal iteration caps; DESIGN.md section 6 holds the measurements.  beta in LLR units. */
int nrldpc_default_rule(int32_t bg, int32_t n_layers, float* alpha, float* beta);

/* Parameter helpers shared with the host-side chain (no device work). */
int nrldpc_set_index(int32_t Z);                        /* get_3gpp_set_index.m:5-11; -1 if invalid  */
int nrldpc_lifting_size(int32_t K_b, int32_t K_prime);  /* get_3gpp_lifting_size.m:5-16; -1 if none */

const char* nrldpc_strerror(int code);
const char* nrldpc_last_error(void); /* text of the most recent failure on this thread */
const char* nrldpc_version(void);
const char* nrldpc_build_id(void);   /* hash of the sources this binary was built from (build.py: source_id) */
const char* nrldpc_kernel_id(void);  /* hash of the decoder kernels' sources alone (build.py: kernel_id): ties a rocprofv3
                                        summary under profiles/ to the kernels it measured */

#ifdef __cplusplus
}
#endif
#endif /* NRLDPC_H */
