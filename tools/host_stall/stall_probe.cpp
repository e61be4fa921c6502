// stall_probe.cpp -- times repeated host-pointer decode calls through the C ABI (include/nrldpc.h), nothing else in the process:
// no Python, no torch, output arrays allocated and touched once.  Used to take apart the alternating 25-35 ms wait of
// nrldpc_decode (byte-per-bit output) that round 4 left open.
//   g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o stall_probe
//   NRLDPC_HOST_TRACE=1 ./stall_probe f16|f32|f64 packed(0|1) reps [batch] [bg] [Z] [n_layers] [zero_from_col] [early_term]
//   n_layers: 0 all rows, 4.., -1 = NRLDPC_LAYERS_AUTO; zero_from_col: base-graph columns from this one on hold LLR 0 (a rate-matched block)
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nrldpc.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uniform01() {
    g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
    return (double)((g_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static uint16_t f2h(float f) { // round-to-nearest-even float -> half, finite inputs of moderate size only
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e <= 0) return (uint16_t)sign;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    uint32_t h = sign | ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)h;
}
static void cpu_stat(const char* tag) {
    FILE* f = std::fopen("/sys/fs/cgroup/cpu.stat", "r");
    if (!f) return;
    char line[128];
    std::fprintf(stderr, "[cpu.stat %s]", tag);
    while (std::fgets(line, sizeof line, f))
        if (std::strstr(line, "throttled")) { line[std::strcspn(line, "\n")] = 0; std::fprintf(stderr, " %s;", line); }
    std::fprintf(stderr, "\n");
    std::fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: stall_probe f16|f32|f64 packed reps [batch bg Z n_layers]\n"); return 2; }
    const char* dt = argv[1];
    const int packed = std::atoi(argv[2]), reps = std::atoi(argv[3]);
    const int batch = argc > 4 ? std::atoi(argv[4]) : 4096, bg = argc > 5 ? std::atoi(argv[5]) : 1, Z = argc > 6 ? std::atoi(argv[6]) : 384;
    const int nl = argc > 7 ? std::atoi(argv[7]) : 0, zero_from = argc > 8 ? std::atoi(argv[8]) : 1 << 30, et = argc > 9 ? std::atoi(argv[9]) : 0;
    nrldpc_cfg cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.bg = bg; cfg.Z = Z; cfg.max_iter = 25; cfg.early_term = et; cfg.n_layers = nl;
    cfg.llr_dtype = !std::strcmp(dt, "f64") ? NRLDPC_LLR_F64 : !std::strcmp(dt, "f16") ? NRLDPC_LLR_F16 : NRLDPC_LLR_F32;
    nrldpc_handle h = nullptr;
    if (nrldpc_create(&cfg, &h) != NRLDPC_OK) { std::fprintf(stderr, "create: %s\n", nrldpc_last_error()); return 1; }
    nrldpc_dims d; d.struct_size = sizeof d;
    nrldpc_get_dims(h, &d);
    const size_t K = (size_t)d.K, N = (size_t)d.N_cw, n = N * (size_t)batch;
    const size_t es = cfg.llr_dtype == NRLDPC_LLR_F64 ? 8 : cfg.llr_dtype == NRLDPC_LLR_F16 ? 2 : 4;
    std::vector<char> llr(n * es);
    for (size_t i = 0; i < n; ++i) { // all-zero codeword at a comfortable SNR: content is irrelevant to the timing at fixed iterations
        float x = 4.0f + 2.8f * (float)(uniform01() + uniform01() + uniform01() - 1.5);
        if ((int)((i % N) / (size_t)Z) >= zero_from) x = 0.0f;
        if (es == 8) reinterpret_cast<double*>(llr.data())[i] = x;
        else if (es == 4) reinterpret_cast<float*>(llr.data())[i] = x;
        else reinterpret_cast<uint16_t*>(llr.data())[i] = f2h(x);
    }
    std::vector<uint8_t> hard(packed ? ((K + 7) / 8) * (size_t)batch : K * (size_t)batch, 1);
    std::vector<int32_t> it(batch, 0);
    cpu_stat("before");
    for (int r = 0; r < reps; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = packed ? nrldpc_decode_packed(h, llr.data(), batch, hard.data(), nullptr)
                              : nrldpc_decode(h, llr.data(), batch, hard.data(), nullptr, nullptr);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc != NRLDPC_OK) { std::fprintf(stderr, "decode: %s\n", nrldpc_last_error()); return 1; }
        std::printf("call %2d  %s %s  %8.3f ms  %.2f Gbit/s\n", r, dt, packed ? "packed" : "byte-per-bit", ms, batch * (double)K / ms / 1e6);
        std::fflush(stdout);
    }
    cpu_stat("after");
    int32_t used = 0;
    nrldpc_last_layers(h, &used);
    std::printf("layers of the last call: %d\n", used);
    size_t ones = 0;
    for (size_t i = 0; i < (packed ? hard.size() : hard.size()); ++i) ones += hard[i] != 0;
    std::printf("nonzero output bytes: %zu\n", ones);
    nrldpc_destroy(h);
    return 0;
}
