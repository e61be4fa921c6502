// nrldpc_mex.cpp -- MEX gateway between the reference's System objects and libnrldpc_hip.so (include/nrldpc.h).
//
//   mex -I<repo>/include -L<repo>/ldpc-3gpp-matlab_amd -lnrldpc_hip nrldpc_mex.cpp
//
// Replaces, in robmaunder/ldpc-3gpp-matlab (see matlab/ldpc-3gpp-matlab.patch for the edits to the .m files):
//   comm.LDPCDecoder(...) construction      NRLDPCDecoder.m:117-121   -> nrldpc_mex('create', BG, Z_c, iterations)
//   step(obj.hLDPCDecoder, cw_tilde)        NRLDPCDecoder.m:257-266   -> nrldpc_mex('decode', id, cw_tilde)   (all C blocks)
//   comm.LDPCEncoder(...) / step(...)       NRLDPCEncoder.m:49,158    -> nrldpc_mex('create', ...) / nrldpc_mex('encode', id, c)
//   release of those toolbox objects                                  -> nrldpc_mex('destroy', id)
//
// Commands
//   id           = nrldpc_mex('create', BG, Z_c, iterations [, n_layers [, alpha [, beta]]])
//                  n_layers omitted / -1 = NRLDPC_LAYERS_AUTO: every call decodes the rows its own cw_tilde needs -- read off the
//                  data, exact for any rv_id / HARQ state (include/nrldpc.h "Active layers"; at plot_BLER_vs_SNR.m's defaults 21 of
//                  BG2's 42 rows, at R = 8/9 5 of BG1's 46); 0 = every row of H, as the reference decodes (NRLDPCDecoder.m:120);
//                  alpha 0 / omitted = the library's rate-dependent check-node rule (nrldpc_default_rule)
//   [c_hat, it, nl] = nrldpc_mex('decode', id, cw_tilde [, n_layers])   cw_tilde: (N+2*Z_c) x C double OR single, +inf fillers, 0 punctured;
//                                                       n_layers: the count of THIS call ONLY -- the handle keeps its own setting
//                                                       for calls without it (a caller that knows its rate -- E_r, k_0, N_cb of
//                                                       NRLDPC.m:463-543: the patched NRLDPCDecoder.LDPC_coding -- saves the scan AUTO
//                                                       makes: with MATLAB doubles the host side is bound by reading the array)
//                                                       c_hat: K x C double in {0,1}; it: C x 1 int32 iterations run;
//                                                       nl: the layer count the call ran with
//   nrldpc_mex('set_layers', id, n_layers)              the count of the calls that follow (0 all, 4..rows, -1 auto)
//   cw           = nrldpc_mex('encode', id, c)          c: K x C double in {0,1} (no NaN) -> (N+2*Z_c) x C double
//   [a, b]       = nrldpc_mex('default_rule', BG, n_layers)
//   nrldpc_mex('destroy', id)
//   One MATLAB process, several GPUs (nrldpc_pool_*: codeword batches shard with no collective; plot_BLER_vs_SNR.m:23-27
//   runs "parallel instances" by hand instead):
//   pid          = nrldpc_mex('pool_create', BG, Z_c, iterations, device_ids [, chunks_per_device [, n_layers]])   (n_layers as above)
//   [c_hat, it]  = nrldpc_mex('pool_decode', pid, cw_tilde)   any number of columns (double), dealt to the GPUs of the pool
//   nrldpc_mex('pool_destroy', pid)
// Errors carry the reference's two identifiers (NRLDPCDecoder.m:149, NRLDPC.m:240-294): callers that catch
// 'ldpc_3gpp_matlab:UnsupportedParameters' and skip (plot_BLER_vs_SNR.m:173, testbench.m:51) keep working.
//
// This file cannot be built into a MEX file in the build image or on the GPU box (no MATLAB); the CPU suite compile-checks
// it against a stub of the MEX API (tests/mex_stub/mex.h, tests/test_capi_symbols.py), and the same entry points are
// driven the same way -- host pointers, doubles in, one call per batch of columns -- by tests/abi_caller/abi_caller.cpp.
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "mex.h"
#include "nrldpc.h"

namespace {

std::map<uint64_t, nrldpc_handle> g_handles; // registry of live codecs; the MEX file stays locked while it is non-empty
struct Pool { nrldpc_pool_handle p; nrldpc_dims d; };
std::map<uint64_t, Pool> g_pools;             // ... or this one
uint64_t g_next = 1;

void at_exit() {
    for (auto& kv : g_handles) nrldpc_destroy(kv.second);
    g_handles.clear();
    for (auto& kv : g_pools) nrldpc_pool_destroy(kv.second.p);
    g_pools.clear();
}

bool registry_empty() { return g_handles.empty() && g_pools.empty(); }

void check(int rc) {
    if (rc == NRLDPC_OK) return;
    const char* id = (rc == NRLDPC_ERR_UNSUPPORTED) ? "ldpc_3gpp_matlab:UnsupportedParameters" : "ldpc_3gpp_matlab:Error";
    mexErrMsgIdAndTxt(id, "%s", nrldpc_last_error());
}

nrldpc_handle handle_of(const mxArray* a) {
    const uint64_t id = (uint64_t)mxGetScalar(a);
    auto it = g_handles.find(id);
    if (it == g_handles.end()) mexErrMsgIdAndTxt("ldpc_3gpp_matlab:Error", "unknown or released codec handle.");
    return it->second;
}

void need(bool ok, const char* msg) {
    if (!ok) mexErrMsgIdAndTxt("ldpc_3gpp_matlab:Error", "%s", msg);
}

} // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    need(nrhs >= 1 && mxIsChar(prhs[0]), "first argument should be a command string.");
    char cmd[32];
    mxGetString(prhs[0], cmd, sizeof cmd);

    if (!strcmp(cmd, "create")) {
        need(nrhs >= 4, "create needs BG, Z_c and iterations.");
        nrldpc_cfg cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.bg = (int32_t)mxGetScalar(prhs[1]);
        cfg.Z = (int32_t)mxGetScalar(prhs[2]);
        cfg.max_iter = (int32_t)mxGetScalar(prhs[3]);                  // 'MaximumIterationCount', NRLDPCDecoder.m:41,120
        cfg.n_layers = nrhs > 4 ? (int32_t)mxGetScalar(prhs[4]) : NRLDPC_LAYERS_AUTO; // read off each call's cw_tilde; 0: the full H
        cfg.alpha = nrhs > 5 ? (float)mxGetScalar(prhs[5]) : 0.0f;     // 0: rule chosen by the library
        cfg.beta = nrhs > 6 ? (float)mxGetScalar(prhs[6]) : 0.0f;
        cfg.early_term = 1;                                            // 'Parity check satisfied', NRLDPCDecoder.m:120
        cfg.llr_dtype = NRLDPC_LLR_F64;                                // MATLAB doubles straight in
        nrldpc_handle h = nullptr;
        check(nrldpc_create(&cfg, &h));
        if (registry_empty()) { mexLock(); mexAtExit(at_exit); }
        g_handles[g_next] = h;
        plhs[0] = mxCreateDoubleScalar((double)g_next++);
    } else if (!strcmp(cmd, "decode")) {
        need((nrhs == 3 || nrhs == 4) && (mxIsDouble(prhs[2]) || mxIsSingle(prhs[2])) && !mxIsComplex(prhs[2]),
             "decode needs a handle and a real double or single matrix [and a layer count].");
        nrldpc_handle h = handle_of(prhs[1]);
        nrldpc_dims d;
        d.struct_size = sizeof d;
        check(nrldpc_get_dims(h, &d));
        need((int)mxGetM(prhs[2]) == d.N_cw, "cw_tilde should have N+2*Z_c rows.");
        const int C = (int)mxGetN(prhs[2]);
        // the caller's array as it is: a `single` cw_tilde is half the bytes for the copy threads to read
        check(nrldpc_set_llr_dtype(h, mxIsSingle(prhs[2]) ? NRLDPC_LLR_F32 : NRLDPC_LLR_F64));
        // bit-packed over PCIe (nrldpc_decode_packed, ABI revision 4): K/8 bytes per column come back instead of K
        const size_t KB8 = ((size_t)d.K + 7) / 8;
        std::vector<uint8_t> packed(KB8 * (size_t)(C > 0 ? C : 1));
        mxArray* it = mxCreateNumericMatrix(C, 1, mxINT32_CLASS, mxREAL);
        // the 4th argument is the count of THIS call (nrldpc_decode_packed_layers, ABI revision 6): nothing sticks to the handle, so a
        // retransmission that omits it runs under the handle's own setting again ('create' / 'set_layers': AUTO by default)
        if (nrhs == 4) check(nrldpc_decode_packed_layers(h, mxGetData(prhs[2]), C, packed.data(), (int32_t*)mxGetData(it), (int32_t)mxGetScalar(prhs[3])));
        else check(nrldpc_decode_packed(h, mxGetData(prhs[2]), C, packed.data(), (int32_t*)mxGetData(it)));
        plhs[0] = mxCreateDoubleMatrix(d.K, C, mxREAL);                // K x C double {0,1}, what double(step(...)) gives, :265
        double* o = mxGetPr(plhs[0]);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < d.K; ++k) o[(size_t)c * d.K + k] = (double)((packed[(size_t)c * KB8 + (k >> 3)] >> (k & 7)) & 1);
        if (nlhs > 1) plhs[1] = it; else mxDestroyArray(it);
        if (nlhs > 2) {
            int32_t nl = 0;
            check(nrldpc_last_layers(h, &nl));
            plhs[2] = mxCreateDoubleScalar((double)nl);
        }
    } else if (!strcmp(cmd, "set_layers")) {
        need(nrhs == 3, "set_layers needs a handle and a layer count.");
        check(nrldpc_set_layers(handle_of(prhs[1]), (int32_t)mxGetScalar(prhs[2])));
    } else if (!strcmp(cmd, "encode")) {
        need(nrhs == 3 && mxIsDouble(prhs[2]) && !mxIsComplex(prhs[2]), "encode needs a handle and a real double matrix.");
        nrldpc_handle h = handle_of(prhs[1]);
        nrldpc_dims d;
        d.struct_size = sizeof d;
        check(nrldpc_get_dims(h, &d));
        need((int)mxGetM(prhs[2]) == d.K, "c should have K rows.");
        const int C = (int)mxGetN(prhs[2]);
        const double* c = mxGetPr(prhs[2]);
        std::vector<uint8_t> info((size_t)d.K * (size_t)(C > 0 ? C : 1)), cw((size_t)d.N_cw * (size_t)(C > 0 ? C : 1));
        for (size_t i = 0; i < (size_t)d.K * C; ++i) {
            need(c[i] == 0.0 || c[i] == 1.0, "c should hold bits (fillers already set to 0, NRLDPCEncoder.m:153).");
            info[i] = (uint8_t)c[i];
        }
        check(nrldpc_encode(h, info.data(), C, cw.data()));            // systematic [c; w], H*cw = 0, NRLDPCEncoder.m:158
        plhs[0] = mxCreateDoubleMatrix(d.N_cw, C, mxREAL);
        double* o = mxGetPr(plhs[0]);
        for (size_t i = 0; i < (size_t)d.N_cw * C; ++i) o[i] = (double)cw[i];
    } else if (!strcmp(cmd, "default_rule")) {
        need(nrhs >= 2, "default_rule needs BG [, n_layers].");
        float a = 0, b = 0;
        check(nrldpc_default_rule((int32_t)mxGetScalar(prhs[1]), nrhs > 2 ? (int32_t)mxGetScalar(prhs[2]) : 0, &a, &b));
        plhs[0] = mxCreateDoubleScalar(a);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(b);
    } else if (!strcmp(cmd, "destroy")) {
        need(nrhs == 2, "destroy needs a handle.");
        const uint64_t id = (uint64_t)mxGetScalar(prhs[1]);
        auto it = g_handles.find(id);
        if (it != g_handles.end()) {
            nrldpc_destroy(it->second);
            g_handles.erase(it);
            if (registry_empty()) mexUnlock();
        }
    } else if (!strcmp(cmd, "pool_create")) {
        need(nrhs >= 5 && mxIsDouble(prhs[4]), "pool_create needs BG, Z_c, iterations and a vector of device ordinals.");
        nrldpc_cfg cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.bg = (int32_t)mxGetScalar(prhs[1]);
        cfg.Z = (int32_t)mxGetScalar(prhs[2]);
        cfg.max_iter = (int32_t)mxGetScalar(prhs[3]);
        cfg.n_layers = nrhs > 6 ? (int32_t)mxGetScalar(prhs[6]) : NRLDPC_LAYERS_AUTO;
        cfg.early_term = 1;                                            // 'Parity check satisfied', NRLDPCDecoder.m:120
        cfg.llr_dtype = NRLDPC_LLR_F64;
        const size_t nd = mxGetNumberOfElements(prhs[4]);
        need(nd >= 1 && nd <= 64, "between 1 and 64 device ordinals.");
        std::vector<int32_t> ids(nd);
        for (size_t i = 0; i < nd; ++i) ids[i] = (int32_t)mxGetPr(prhs[4])[i];
        const int32_t chunks = nrhs > 5 ? (int32_t)mxGetScalar(prhs[5]) : 3;
        Pool pl;
        pl.p = nullptr;
        check(nrldpc_pool_create(&cfg, ids.data(), (int32_t)nd, chunks, &pl.p));
        // dimensions of the code: from a throw-away handle on the first device of the pool
        cfg.device_id = ids[0];
        nrldpc_handle h = nullptr;
        int rc = nrldpc_create(&cfg, &h);
        if (rc == NRLDPC_OK) { pl.d.struct_size = sizeof pl.d; rc = nrldpc_get_dims(h, &pl.d); nrldpc_destroy(h); }
        if (rc != NRLDPC_OK) { nrldpc_pool_destroy(pl.p); check(rc); }
        if (registry_empty()) { mexLock(); mexAtExit(at_exit); }
        g_pools[g_next] = pl;
        plhs[0] = mxCreateDoubleScalar((double)g_next++);
    } else if (!strcmp(cmd, "pool_decode")) {
        need(nrhs == 3 && mxIsDouble(prhs[2]) && !mxIsComplex(prhs[2]), "pool_decode needs a pool id and a real double matrix.");
        auto pit = g_pools.find((uint64_t)mxGetScalar(prhs[1]));
        need(pit != g_pools.end(), "unknown or released pool.");
        const nrldpc_dims& d = pit->second.d;
        need((int)mxGetM(prhs[2]) == d.N_cw, "cw_tilde should have N+2*Z_c rows.");
        const int C = (int)mxGetN(prhs[2]);
        const size_t KB8 = ((size_t)d.K + 7) / 8;                      // bit-packed over PCIe, as 'decode'
        std::vector<uint8_t> packed(KB8 * (size_t)(C > 0 ? C : 1));
        mxArray* it = mxCreateNumericMatrix(C, 1, mxINT32_CLASS, mxREAL);
        check(nrldpc_pool_decode_packed(pit->second.p, mxGetPr(prhs[2]), C, packed.data(), (int32_t*)mxGetData(it)));
        plhs[0] = mxCreateDoubleMatrix(d.K, C, mxREAL);
        double* o = mxGetPr(plhs[0]);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < d.K; ++k) o[(size_t)c * d.K + k] = (double)((packed[(size_t)c * KB8 + (k >> 3)] >> (k & 7)) & 1);
        if (nlhs > 1) plhs[1] = it; else mxDestroyArray(it);
    } else if (!strcmp(cmd, "pool_destroy")) {
        need(nrhs == 2, "pool_destroy needs a pool id.");
        auto pit = g_pools.find((uint64_t)mxGetScalar(prhs[1]));
        if (pit != g_pools.end()) {
            nrldpc_pool_destroy(pit->second.p);
            g_pools.erase(pit);
            if (registry_empty()) mexUnlock();
        }
    } else {
        mexErrMsgIdAndTxt("ldpc_3gpp_matlab:Error", "unknown command '%s'.", cmd);
    }
}
