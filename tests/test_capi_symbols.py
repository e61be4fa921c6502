"""The C-ABI shared object loads and exports every function include/nrldpc.h declares.
No compute calls here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "nrldpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nrldpc_[a-z_0-9]+)\s*\(", txt)))


def test_header_lists_the_boundary():
    names = declared_functions()
    for must in ("nrldpc_create", "nrldpc_decode", "nrldpc_decode_dev", "nrldpc_encode", "nrldpc_destroy",
                 "nrldpc_strerror", "nrldpc_rate_recover_dev", "nrldpc_crc_check_dev", "nrldpc_crc_attach_dev",
                 "nrldpc_rate_match_dev"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert set(pkg._capi.EXPORTS) <= set(declared_functions())
    assert b"gfx950" in lib.nrldpc_version()


def test_no_torch_types_in_signatures():
    txt = open(os.path.join(ROOT, "include", "nrldpc.h")).read()
    assert "torch" not in txt and "at::" not in txt and "hipStream_t stream" not in txt


def test_helpers_without_device(pkg):
    lib = pkg.load()
    assert lib.nrldpc_set_index(384) == 1 and lib.nrldpc_set_index(17) == -1
    assert lib.nrldpc_lifting_size(22, 8448) == 384
    assert lib.nrldpc_strerror(1) == b"unsupported parameters"


def test_host_quantiser_is_the_kernels_ingest(pkg):
    """nrldpc_quantise_llr (what the host path puts on the wire) against a numpy restatement of the decoder kernels'
    ingest(): f32 multiply by the scale, NaN -> 0, clamp to +-127, round to nearest even; +inf -> -128; -inf flagged.
    All three boundary dtypes, every scale, sizes around the vector width."""
    import numpy as np
    rng = np.random.default_rng(5)
    for dt in (np.float64, np.float32, np.float16):
        for scale in (1, 2, 4, 8, 16, 32):
            for n in (1, 7, 8, 33, 1000, 4099):
                x = (rng.standard_normal(n) * rng.choice([0.05, 1.0, 6.0, 40.0])).astype(dt)
                k = rng.integers(0, n, 6)
                x[k[0]] = np.nan; x[k[1]] = np.inf; x[k[2]] = (rng.integers(-300, 300) + 0.5) / scale
                x[k[3]] = 127.5 / scale; x[k[4]] = -0.0
                neg = bool(n > 8 and rng.integers(0, 2))
                if neg:
                    x[k[5]] = -np.inf
                q, flag = pkg._capi.quantise_llr(x, scale)
                for path in ("0", "1", "2"):  # plain C++, AVX2 + F16C, AVX-512 (a path the CPU lacks falls to the next one down)
                    os.environ["NRLDPC_HOST_QUANT_PATH"] = path
                    try:
                        qp, fp = pkg._capi.quantise_llr(x, scale)
                    finally:
                        del os.environ["NRLDPC_HOST_QUANT_PATH"]
                    assert (qp == q).all() and fp == flag, (dt, scale, n, path)
                x32 = x.astype(np.float32)
                with np.errstate(invalid="ignore", over="ignore"):
                    y = x32 * np.float32(scale)
                y = np.where(np.isnan(y), np.float32(0), y)
                ref = np.rint(np.clip(y, -127, 127)).astype(np.int8)
                ref[x32 == np.inf] = -128
                assert (q == ref).all(), (dt, scale, n)
                assert flag == bool((x32 == -np.inf).any())


def test_create_rejects_bad_parameters_before_touching_the_device(pkg):
    C = pkg._capi
    lib = pkg.load()
    h = ctypes.c_void_p()
    for bg, Z, nl, it in ((3, 384, 0, 10), (1, 17, 0, 10), (1, 384, 0, 0), (1, 384, 0, 5000)):
        cfg = C.Cfg(bg, Z, nl, it, 1, 0.0, 0, 0, 0, 0)
        assert lib.nrldpc_create(ctypes.byref(cfg), ctypes.byref(h)) == C.ERR_UNSUPPORTED
        assert h.value is None
    cfg = C.Cfg(1, 384, 3, 10, 1, 0.0, 0, 0, 0, 0)  # fewer than the 4 core layers
    assert lib.nrldpc_create(ctypes.byref(cfg), ctypes.byref(h)) == C.ERR_UNSUPPORTED
    with pytest.raises(pkg.UnsupportedParameters):
        pkg.Codec(1, 100)


def test_product_never_imports_the_oracle():
    """Guard for the rule that only tests/, smoke() and bench's cpu_baseline may touch oracle/."""
    pk = os.path.join(ROOT, "ldpc-3gpp-matlab_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "oracle/" not in src.replace("oracle/nrldpc_oracle.c", ""), f


def test_matlab_patch_applies(tmp_path):
    """matlab/ldpc-3gpp-matlab.patch against the reference tree (build box only: /root/reference is not on the GPU box)."""
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(ref) or not shutil.which("patch"):
        pytest.skip("reference tree or patch(1) not available")
    for f in ("NRLDPCDecoder.m", "NRLDPCEncoder.m"):
        shutil.copy(os.path.join(ref, f), tmp_path / f)
    patch = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matlab", "ldpc-3gpp-matlab.patch")
    subprocess.check_call(["patch", "-p1", "--binary", "-s", "-i", patch], cwd=tmp_path)
    dec = (tmp_path / "NRLDPCDecoder.m").read_text(encoding="latin-1")
    assert "nrldpc_mex('decode', obj.hLDPCDecoder, cw_tilde, n_layers)" in dec and "comm.LDPCDecoder(" not in dec
    # the exact active-row count goes with every call: derived from E_r / k_0 / N_cb, sticky while HARQ state is
    # pending, cleared by setupImpl and resetImpl (tests/test_testbench.py checks the arithmetic against NRLDPC_LAYERS_AUTO)
    assert dec.count("function n_layers = active_layers(obj)") == 1 and dec.count("obj.layers_seen = 0;") == 2
    assert "nrldpc_mex('encode'" in (tmp_path / "NRLDPCEncoder.m").read_text(encoding="latin-1")
    enc = (tmp_path / "NRLDPCEncoder.m").read_text(encoding="latin-1")
    assert enc.count("releaseImpl") == 1 and dec.count("function releaseImpl") == 1  # both objects give their codec back


def test_mex_gateway_compiles_against_the_stub_mex_api():
    """matlab/nrldpc_mex.cpp cannot be built into a MEX file here (no MATLAB); it is compile-checked against
    tests/mex_stub/mex.h (declarations of the documented MEX API calls it uses) with warnings as errors, so that a
    change of include/nrldpc.h that breaks the gateway is caught on the CPU."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "tests", "mex_stub"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "matlab", "nrldpc_mex.cpp")])
    src = open(os.path.join(ROOT, "matlab", "nrldpc_mex.cpp")).read()
    for cmd in ("create", "decode", "set_layers", "encode", "destroy", "default_rule", "pool_create", "pool_decode", "pool_destroy"):
        assert '"%s"' % cmd in src, cmd


def test_abi_revision_and_struct_size_guard(pkg):
    """ABI revision 3 (kept by revisions 4 and 5, which add entry points and a value of n_layers, not fields): nrldpc_cfg /
    nrldpc_dims carry their size; a caller built against another revision is refused instead of having memory past its struct
    read or written."""
    C = pkg._capi
    lib = pkg.load()
    assert lib.nrldpc_abi_version() == C.ABI_VERSION == 6
    hdr = open(os.path.join(ROOT, "include", "nrldpc.h")).read()
    assert "#define NRLDPC_ABI_VERSION 6" in hdr and "#define NRLDPC_LAYERS_AUTO (-1)" in hdr
    cfg = C.Cfg(1, 384, 0, 10, 1, 0.0, 0, 0, 0, 0)
    assert cfg.struct_size == ctypes.sizeof(C.Cfg)
    cfg.struct_size = ctypes.sizeof(C.Cfg) - 4  # the r1 layout (no beta)
    h = ctypes.c_void_p()
    assert lib.nrldpc_create(ctypes.byref(cfg), ctypes.byref(h)) == C.ERR_ARG and h.value is None
    assert b"struct_size" in lib.nrldpc_last_error()


def test_count_layers_is_the_definition_of_auto(pkg):
    """nrldpc_count_layers (host function, no device): NRLDPC_LAYERS_AUTO's rule -- n = max(4, c - kb + 1) for the highest
    base-graph column c holding anything but +-0 / NaN in any codeword of the call -- against numpy, for the three boundary
    dtypes, odd lifting sizes (unaligned blocks) and the values that must not count (NaN, -0) or must (+-inf, denormals)."""
    import numpy as np
    C = pkg._capi
    rng = np.random.default_rng(21)
    for bg, (rows, cols, kb) in ((1, (46, 68, 22)), (2, (42, 52, 10))):
        for Z in (2, 7, 15, 64, 208, 384):
            for dt in (np.float64, np.float32, np.float16):
                for trial in range(6):
                    B = int(rng.integers(1, 9))
                    x = np.zeros((B, cols, Z), dt)
                    top = int(rng.integers(0, cols))
                    x[:, : top + 1] = rng.standard_normal((B, top + 1, Z)).astype(dt)
                    if trial == 1:    # only ONE value in the top block of ONE codeword, the last element
                        x[:, top] = 0; x[B - 1, top, Z - 1] = np.finfo(dt).tiny
                    if trial == 2 and top + 2 < cols:   # things that do not count, above the top
                        x[0, top + 1, 0] = np.nan; x[B - 1, top + 2, Z - 1] = -0.0
                    if trial == 3 and top + 1 < cols:   # ... and one that does
                        top += 1; x[B // 2, top, Z // 2] = -np.inf
                    if trial == 4:
                        x[:] = 0; top = 0
                    nz = ((x != 0) & ~np.isnan(x)).any(axis=(0, 2))
                    want = max(4, (int(np.nonzero(nz)[0].max()) if nz.any() else 0) - kb + 1)
                    assert C.count_layers(bg, Z, x.reshape(B, -1)) == want, (bg, Z, dt, trial, top)
    assert pkg.load().nrldpc_count_layers(3, 384, None, 0, 0) == -1 and pkg.load().nrldpc_count_layers(1, 100, None, 0, 0) == -1
    assert C.count_layers(1, 384, np.zeros((0, 68 * 384), np.float32)) == 4


def test_committed_profile_belongs_to_the_tree_kernels(pkg):
    """bench.py takes its VALU instruction counts and HBM traffic from profiles/r06_* only when their nrldpc_kernel_id equals
    the loaded library's (a kernel change without a profile refresh silently falsified a fraction).  This test
    makes the refresh hard to forget: the committed summaries must be those of the decoder kernels in the tree."""
    import json
    kid = pkg._capi._build.kernel_id()
    assert pkg.load().nrldpc_kernel_id().decode() == kid
    for f in ("r06_bench_pmc_summary.json", "r06_traffic_bytes_per_launch.json", "r06_headline_isa_mix.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", f)))
        got = d.get("_nrldpc_kernel_id") or d.get("nrldpc_kernel_id")
        assert got == kid, "%s was measured on kernels %s, the tree holds %s: re-run tools/final_session.sh" % (f, got, kid)
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
    assert line["roofline"]["profile"]["nrldpc_kernel_id"] == kid and line["roofline"]["frac"] is not None
    assert line["roofline"]["frac"] <= 1.0 and line["roofline"]["bound"] == "valu_issue"


def test_kernel_lists_of_the_build_and_of_the_dispatch_agree():
    """build.py compiles one translation unit per (BG, Z[, layers]) of nrldpc_kernels.h's X-macro lists, which are also what
    nrldpc_decode.hip dispatches on: the two must name the same pairs (a missing unit is a link error, a superfluous one dead
    code), and the packed-geometry exceptions must be members of the packed list."""
    import importlib
    bld = importlib.import_module("ldpc-3gpp-matlab_amd.build")
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldpc-3gpp-matlab_amd", "csrc",
                            "nrldpc_kernels.h")).read().replace("\\\n", " ")
    src += open(os.path.join(ROOT, "ldpc-3gpp-matlab_amd", "csrc", "nrldpc_dispatch_lists.h")).read().replace("\\\n", " ")

    def pairs(name):
        m = re.search(r"#define %s\(X\)([^\n]*)" % name, src)
        assert m, name
        return [tuple(int(x) for x in t.split(",")) for t in re.findall(r"X\(([^)]*)\)", m.group(1))]

    assert sorted(pairs("NRLDPC_Z64_LIST")) == sorted(bld.Z64_PAIRS)
    assert sorted(pairs("NRLDPC_Z64P_LIST")) == sorted(bld.Z64P_PAIRS)
    assert sorted(pairs("NRLDPC_Z64PR_LIST")) == sorted(bld.Z64PR)
    assert sorted(pairs("NRLDPC_Z64_NL_LIST")) == sorted(bld.Z64_NL)
    assert sorted(pairs("NRLDPC_Z64P_NL_LIST")) == sorted(bld.Z64P_NL)
    assert sorted(pairs("NRLDPC_Z64I_LIST")) == sorted(bld.Z64I)
    assert set(pairs("NRLDPC_Z64P_NOT_ET")) <= set(bld.Z64P_PAIRS)
    # a packed size that has no block-geometry unit falls back to the run-time-Z kernel for pruned rows / soft output: fine;
    # but every lifting size of TS 38.212 must be a legal Z for it
    all_z = sorted(a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384)
    assert all(z in all_z for _, z in bld.Z64_PAIRS + bld.Z64P_PAIRS + [(b, z) for b, z, _ in bld.Z64PR])
    # interleaved block geometry: the virtual size is a block-geometry shape of full (or nearly full) waves, at most 8 per half
    # (at most 8 waves of at most 64 rows per half, and a mode that serves something)
    assert all(z in all_z and 1 <= n and z * n <= 512 and 1 <= m <= 7 for _, z, n, m in bld.Z64I)
    assert len({(b, z) for b, z, _, _ in bld.Z64I}) == len(bld.Z64I)
    for b, z, n, m in bld.Z64I:  # the workgroup shape z64_blk / Z64P derive from the virtual size (nrldpc_decode_z64.h, _z64p.h)
        zc = z * n
        blk = 64 if zc % 64 == 0 else max(d for d in range(4, 64) if zc % d == 0)
        waves = zc // blk
        assert 2 * waves <= 16 and blk >= 40, (b, z, n)  # at most 1024 threads; no shape with more than 3/8 of the lanes idle
        nc, ext = (26, 42) if b == 1 else (14, 38)
        image = nc * (256 + (zc + 64) * 4) + 256 + 4 * ((n + 1 + 3) // 4 * 4)
        assert image <= 160 * 1024 and nc * (256 + (zc + 64) * 4) < 65536, (b, z, n)  # LDS budget; 16-bit LDS immediates
        # the flag words are cleared by raw thread id in one pass, and the lanes >= blk of every wave have retired by then
        # (the kernel's own static_assert)
        assert n + 1 <= (2 * waves * 64 if blk == 64 else blk), (b, z, n)
