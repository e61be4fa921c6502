// abi_caller.cpp -- a plain C++ caller of include/nrldpc.h, the way a MEX gateway (matlab/nrldpc_mex.cpp) reaches the
// library: host pointers only, MATLAB-style column-major doubles, one call per batch of code blocks, error codes.
// Since round 5 it makes the gateway's REAL calls: nrldpc_decode_packed on doubles and singles, NRLDPC_LAYERS_AUTO against explicit
// counts, nrldpc_pool_* with two logical shards, and the struct-size guard against a revision-3 caller.
// No HIP headers, no Python.  Built by tests/test_abi_caller_gpu.py:
//     g++ -O2 -std=c++17 -I include tests/abi_caller/abi_caller.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o abi_caller
// usage: abi_caller <bg> <Z> <C> <EsN0_dB> <iterations> [n_filler]
// Prints one JSON line; exit code 0 iff every check held.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "nrldpc.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uniform01() { // xorshift64*: the caller's own noise, nothing from the library
    g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
    return (double)((g_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double gauss() { return std::sqrt(-2.0 * std::log(uniform01() + 1e-300)) * std::cos(6.283185307179586 * uniform01()); }

#define CHECK(cond, what)                                                         \
    do {                                                                          \
        if (!(cond)) { std::fprintf(stderr, "FAILED: %s (%s)\n", what, nrldpc_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: abi_caller bg Z C EsN0_dB iterations [n_filler]\n"); return 2; }
    const int bg = std::atoi(argv[1]), Z = std::atoi(argv[2]), C = std::atoi(argv[3]), iters = std::atoi(argv[5]);
    const double esn0 = std::atof(argv[4]);
    const int n_filler = argc > 6 ? std::atoi(argv[6]) : 0;

    // error convention first: an invalid lifting size is "unsupported", not a crash (get_3gpp_set_index.m:10)
    nrldpc_cfg bad;
    std::memset(&bad, 0, sizeof bad);
    bad.struct_size = sizeof bad;
    bad.bg = bg; bad.Z = 100; bad.max_iter = iters; bad.llr_dtype = NRLDPC_LLR_F64;
    nrldpc_handle h = nullptr;
    CHECK(nrldpc_create(&bad, &h) == NRLDPC_ERR_UNSUPPORTED && h == nullptr, "invalid Z is NRLDPC_ERR_UNSUPPORTED");
    CHECK(nrldpc_create(nullptr, &h) == NRLDPC_ERR_ARG, "null cfg is NRLDPC_ERR_ARG");

    nrldpc_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.bg = bg; cfg.Z = Z; cfg.max_iter = iters; cfg.early_term = 1;   // NRLDPCDecoder.m:120
    CHECK(nrldpc_create(&cfg, &h) == NRLDPC_ERR_ARG && h == nullptr, "a cfg without struct_size (other ABI revision) is refused");
    cfg.struct_size = sizeof cfg;
    cfg.llr_dtype = NRLDPC_LLR_F64;                                     // MATLAB doubles; alpha = 0: the library's rule
    CHECK(nrldpc_create(&cfg, &h) == NRLDPC_OK, "nrldpc_create");
    nrldpc_dims d;
    d.struct_size = sizeof d;
    CHECK(nrldpc_abi_version() == NRLDPC_ABI_VERSION, "header and library are the same ABI revision");
    CHECK(nrldpc_get_dims(h, &d) == NRLDPC_OK, "nrldpc_get_dims");
    float ra = 0, rb = 0;
    CHECK(nrldpc_default_rule(bg, 0, &ra, &rb) == NRLDPC_OK && ra == d.alpha && rb == d.beta, "create applied nrldpc_default_rule");
    const size_t K = (size_t)d.K, N = (size_t)d.N_cw;

    std::vector<uint8_t> info(K * C), cw(N * C), hard(K * C, 2);
    for (size_t i = 0; i < info.size(); ++i) info[i] = uniform01() < 0.5;
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < n_filler; ++k) info[(size_t)c * K + K - 1 - k] = 0;   // filler bits are zeros (NRLDPCEncoder.m:153)
    CHECK(nrldpc_encode(h, info.data(), C, cw.data()) == NRLDPC_OK, "nrldpc_encode");
    for (size_t i = 0; i < K * C; ++i) CHECK(cw[(i / K) * N + (i % K)] == info[i], "systematic codeword");

    // QPSK over AWGN as plot_BLER_vs_SNR.m:105-106 with NRDemodulator's exact LLRs; first 2Z columns punctured (:262)
    const double mu = 2.0 * std::pow(10.0, esn0 / 10.0);
    std::vector<double> llr(N * C);
    for (int c = 0; c < C; ++c)
        for (size_t v = 0; v < N; ++v) {
            double x = (1.0 - 2.0 * cw[(size_t)c * N + v]) * mu + std::sqrt(2.0 * mu) * gauss();
            if (v < (size_t)(2 * Z)) x = 0.0;
            if (v >= K - (size_t)n_filler && v < K) x = std::numeric_limits<double>::infinity(); // NRLDPCDecoder.m:264
            llr[(size_t)c * N + v] = x;
        }
    std::vector<int32_t> it(C, -1);
    CHECK(nrldpc_decode(h, llr.data(), C, hard.data(), it.data(), nullptr) == NRLDPC_OK, "nrldpc_decode");
    int block_errors = 0, it_max = 0;
    for (int c = 0; c < C; ++c) {
        block_errors += std::memcmp(&hard[(size_t)c * K], &info[(size_t)c * K], K) != 0;
        CHECK(it[c] >= 1 && it[c] <= iters, "iteration counts in range");
        it_max = it[c] > it_max ? it[c] : it_max;
    }
    // one code block per call, the reference's own call pattern (NRLDPCDecoder.m:257-266): same bits
    std::vector<uint8_t> one(K);
    CHECK(nrldpc_decode(h, llr.data(), 1, one.data(), nullptr, nullptr) == NRLDPC_OK, "single-column decode");
    CHECK(std::memcmp(one.data(), hard.data(), K) == 0, "single-column decode equals column 0 of the batch");
    CHECK(nrldpc_decode(h, llr.data(), -1, hard.data(), nullptr, nullptr) == NRLDPC_ERR_ARG, "negative batch is NRLDPC_ERR_ARG");
    CHECK(nrldpc_decode(h, nullptr, 1, hard.data(), nullptr, nullptr) == NRLDPC_ERR_ARG, "null llr is NRLDPC_ERR_ARG");

    // what matlab/nrldpc_mex.cpp 'decode' really calls since ABI revision 4: bit-packed output, doubles in -- same bits, same counts
    const size_t KB8 = (K + 7) / 8;
    std::vector<uint8_t> packed(KB8 * C, 0xff);
    std::vector<int32_t> it2(C, -1);
    CHECK(nrldpc_decode_packed(h, llr.data(), C, packed.data(), it2.data()) == NRLDPC_OK, "nrldpc_decode_packed");
    for (int c = 0; c < C; ++c) {
        CHECK(it2[c] == it[c], "packed call: same iteration counts");
        for (size_t k = 0; k < K; ++k)
            CHECK(((packed[(size_t)c * KB8 + (k >> 3)] >> (k & 7)) & 1) == hard[(size_t)c * K + k], "packed call: same bits");
        if (K % 8) CHECK((packed[(size_t)c * KB8 + KB8 - 1] >> (K % 8)) == 0, "unused bits of the last byte are zero");
    }
    // ... from a `single` array through the same handle (nrldpc_set_llr_dtype): the gateway hands over what the caller holds
    std::vector<float> llr32(llr.begin(), llr.end());
    std::vector<uint8_t> packed32(KB8 * C, 0xff);
    CHECK(nrldpc_set_llr_dtype(h, NRLDPC_LLR_F32) == NRLDPC_OK, "nrldpc_set_llr_dtype");
    CHECK(nrldpc_decode_packed(h, llr32.data(), C, packed32.data(), nullptr) == NRLDPC_OK, "nrldpc_decode_packed on singles");
    CHECK(nrldpc_set_llr_dtype(h, NRLDPC_LLR_F64) == NRLDPC_OK && nrldpc_set_llr_dtype(h, 7) == NRLDPC_ERR_UNSUPPORTED, "llr_dtype back / invalid");
    // (doubles are narrowed to float before anything else happens to them, so the two calls see the same values)
    CHECK(std::memcmp(packed32.data(), packed.data(), packed.size()) == 0, "singles decode like the doubles they came from");

    // ABI revision 5: the layer count belongs to the call.  The gateway creates its handles under NRLDPC_LAYERS_AUTO; a rate-matched
    // block (here: everything above `act` rows not transmitted = LLR 0, NRLDPCDecoder.m:216-234) then decodes `act` rows -- the
    // bits and counts of an explicit nrldpc_set_layers(act), which is what a handle CREATED with n_layers = act gives
    const int rows = d.nrows, act = rows > 12 ? 12 : rows;
    std::vector<double> rm(llr);
    for (int c = 0; c < C; ++c)
        for (size_t v = (size_t)(d.kb + act) * Z; v < N; ++v) rm[(size_t)c * N + v] = 0.0;
    CHECK(nrldpc_count_layers(bg, Z, rm.data(), C, NRLDPC_LLR_F64) == act, "nrldpc_count_layers");
    CHECK(nrldpc_count_layers(bg, Z, llr.data(), C, NRLDPC_LLR_F64) == rows, "nrldpc_count_layers on the full-rate block");
    CHECK(nrldpc_set_layers(h, NRLDPC_LAYERS_AUTO) == NRLDPC_OK, "nrldpc_set_layers(AUTO)");
    std::vector<uint8_t> h_auto(K * C, 2), h_exp(K * C, 3), h_new(K * C, 4);
    std::vector<int32_t> it_auto(C, -1), it_exp(C, -2), it_new(C, -3);
    int32_t nl = 0;
    CHECK(nrldpc_decode(h, rm.data(), C, h_auto.data(), it_auto.data(), nullptr) == NRLDPC_OK, "decode under AUTO");
    CHECK(nrldpc_last_layers(h, &nl) == NRLDPC_OK && nl == act, "AUTO found the active rows");
    CHECK(nrldpc_decode(h, llr.data(), C, hard.data(), nullptr, nullptr) == NRLDPC_OK && nrldpc_last_layers(h, &nl) == NRLDPC_OK && nl == rows,
          "the next call finds its own count");
    CHECK(nrldpc_set_layers(h, act) == NRLDPC_OK, "nrldpc_set_layers(act)");
    CHECK(nrldpc_decode(h, rm.data(), C, h_exp.data(), it_exp.data(), nullptr) == NRLDPC_OK, "decode with an explicit count");
    nrldpc_cfg cfg2 = cfg;
    cfg2.n_layers = act;
    nrldpc_handle h2 = nullptr;
    CHECK(nrldpc_create(&cfg2, &h2) == NRLDPC_OK, "create with n_layers");
    CHECK(nrldpc_decode(h2, rm.data(), C, h_new.data(), it_new.data(), nullptr) == NRLDPC_OK, "decode on that handle");
    nrldpc_destroy(h2);
    CHECK(h_auto == h_exp && h_auto == h_new && it_auto == it_exp && it_auto == it_new, "AUTO == set_layers == create(n_layers), bits and counts");
    CHECK(nrldpc_set_layers(h, 3) == NRLDPC_ERR_UNSUPPORTED && nrldpc_set_layers(h, rows + 1) == NRLDPC_ERR_UNSUPPORTED, "invalid counts are refused");
    // ABI revision 6: the count as an ARGUMENT of one call (what the gateway's 'decode' with a 4th argument calls): the handle, set to
    // every row here, keeps that setting -- a retransmission that omits the count is decoded with all rows again
    CHECK(nrldpc_set_layers(h, NRLDPC_LAYERS_ALL) == NRLDPC_OK, "handle back to every row");
    std::vector<uint8_t> pk_call(KB8 * C, 0xee), pk_all(KB8 * C, 0xdd);
    std::vector<int32_t> it_call(C, -5);
    CHECK(nrldpc_decode_packed_layers(h, rm.data(), C, pk_call.data(), it_call.data(), act) == NRLDPC_OK, "nrldpc_decode_packed_layers");
    CHECK(nrldpc_last_layers(h, &nl) == NRLDPC_OK && nl == act && it_call == it_exp, "the call ran with its own count");
    for (int c = 0; c < C; ++c)
        for (size_t k = 0; k < K; ++k)
            CHECK(((pk_call[(size_t)c * KB8 + (k >> 3)] >> (k & 7)) & 1) == h_exp[(size_t)c * K + k], "per-call count: the bits of set_layers(act)");
    CHECK(nrldpc_decode_packed(h, llr.data(), C, pk_all.data(), nullptr) == NRLDPC_OK && nrldpc_last_layers(h, &nl) == NRLDPC_OK && nl == rows,
          "the next call without a count runs under the handle's own setting: nothing stuck");
    CHECK(std::memcmp(pk_all.data(), packed.data(), packed.size()) == 0, "... and gives the all-rows bits");
    CHECK(nrldpc_decode_packed_layers(h, rm.data(), C, pk_call.data(), nullptr, NRLDPC_LAYERS_AUTO) == NRLDPC_OK && nrldpc_last_layers(h, &nl) == NRLDPC_OK && nl == act,
          "per-call AUTO");
    CHECK(nrldpc_decode_packed_layers(h, rm.data(), C, pk_call.data(), nullptr, 3) == NRLDPC_ERR_UNSUPPORTED, "an invalid per-call count is refused");

    // a caller built against ABI revision 3 (nrldpc_cfg without the crc_* tail) is refused, not misread
    nrldpc_cfg old = cfg;
    old.struct_size = (uint32_t)(sizeof cfg - 3 * sizeof(int32_t));
    CHECK(nrldpc_create(&old, &h2) == NRLDPC_ERR_ARG && h2 == nullptr, "a revision-3-sized nrldpc_cfg is refused");
    nrldpc_pool_handle pool = nullptr;
    const int32_t ids[2] = {0, 0};
    CHECK(nrldpc_pool_create(&old, ids, 2, 2, &pool) == NRLDPC_ERR_ARG && pool == nullptr, "... by nrldpc_pool_create too");

    // the multi-GPU form the gateway's 'pool_*' commands use, two logical shards on device 0: same bits as the handle
    cfg2 = cfg;
    cfg2.n_layers = NRLDPC_LAYERS_AUTO;
    CHECK(nrldpc_pool_create(&cfg2, ids, 2, 2, &pool) == NRLDPC_OK && nrldpc_pool_size(pool) == 2, "nrldpc_pool_create");
    std::vector<uint8_t> p_hard(K * C, 5), p_packed(KB8 * C, 0xff);
    std::vector<int32_t> p_it(C, -4), split(2, -1);
    CHECK(nrldpc_pool_decode(pool, rm.data(), C, p_hard.data(), p_it.data()) == NRLDPC_OK, "nrldpc_pool_decode");
    CHECK(p_hard == h_auto && p_it == it_auto, "pool == handle (AUTO found once for the whole batch)");
    CHECK(nrldpc_pool_last_split(pool, split.data()) == NRLDPC_OK && split[0] + split[1] == C, "every codeword decoded once");
    CHECK(nrldpc_pool_decode_packed(pool, llr.data(), C, p_packed.data(), p_it.data()) == NRLDPC_OK, "nrldpc_pool_decode_packed");
    CHECK(std::memcmp(p_packed.data(), packed.data(), packed.size()) == 0 && p_it == it2, "pool, bit-packed == handle, bit-packed");
    CHECK(nrldpc_pool_set_layers(pool, 0) == NRLDPC_OK && nrldpc_pool_set_layers(pool, 2) == NRLDPC_ERR_UNSUPPORTED, "nrldpc_pool_set_layers");
    // ABI revision 6: per-shard kernel times of a pool (event pairs on the shards' own streams)
    float pms[2] = {-1.0f, -1.0f};
    CHECK(nrldpc_pool_last_kernel_ms(pool, pms) == NRLDPC_ERR_ARG, "no pool timing before it is enabled");
    CHECK(nrldpc_pool_set_timing(pool, 1) == NRLDPC_OK, "nrldpc_pool_set_timing");
    CHECK(nrldpc_pool_decode(pool, llr.data(), C, p_hard.data(), p_it.data()) == NRLDPC_OK, "a timed pool call");
    CHECK(nrldpc_pool_last_kernel_ms(pool, pms) == NRLDPC_OK && pms[0] >= 0.0f && pms[1] >= 0.0f && pms[0] + pms[1] > 0.0f, "nrldpc_pool_last_kernel_ms");
    CHECK(nrldpc_pool_set_timing(pool, 0) == NRLDPC_OK, "pool timing off");
    nrldpc_pool_destroy(pool);
    nrldpc_destroy(h);
    std::printf("{\"bg\": %d, \"Z\": %d, \"C\": %d, \"EsN0_dB\": %.2f, \"block_errors\": %d, \"max_iterations\": %d, "
                "\"alpha\": %.4f, \"beta\": %.4f, \"version\": \"%s\", \"build\": \"%s\"}\n",
                bg, Z, C, esn0, block_errors, it_max, d.alpha, d.beta, nrldpc_version(), nrldpc_build_id());
    return block_errors == 0 ? 0 : 1;
}
