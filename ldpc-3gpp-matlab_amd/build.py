"""Build recipe for libnrldpc_hip.so (gfx950 only; hipcc cross-compiles without a GPU).

Every translation unit is compiled to an object in parallel (the compile-time-Z decoder is instantiated
once per (BG, Z) pair from one source with -D flags), then linked into one in-tree shared object.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
# NRLDPC_BUILD_AB=1: the A/B build -- both decoder forms (one / two threads per row) for every (BG, Z), chosen at run time by
# NRLDPC_SPLIT=0/1 -- into its own library and object directory; load it with NRLDPC_LIB=<path> (tools/bench_all_z.py / tools/bench_configs.py with NRLDPC_SPLIT=0/1 and OUT_SUFFIX)
AB = bool(os.environ.get("NRLDPC_BUILD_AB"))
# NRLDPC_BUILD_ALLMODES=1: the experiment build in which EVERY interleaved entry is compiled for, and serves, all three modes (fixed
# iterations, parity stop with all rows, parity stop with pruned rows) -- what tools/ab_ilv.py measures the list's mode bits from;
# its own library and object directory (seed the directory with a copy of build/: only the changed units recompile)
ALLMODES = bool(os.environ.get("NRLDPC_BUILD_ALLMODES"))
# NRLDPC_BUILD_EXTRA_FLAGS="<flags>": the experiment build with extra compiler flags for every unit (e.g. LLVM scheduling options), into its
# own library and object directory (libnrldpc_hip_x.so, build_x/); load it with NRLDPC_LIB=<path>
XFLAGS = os.environ.get("NRLDPC_BUILD_EXTRA_FLAGS", "").split()
LIB = os.environ.get("NRLDPC_LIB") or os.path.join(HERE, "libnrldpc_hip_ab.so" if AB else "libnrldpc_hip_allmodes.so" if ALLMODES else "libnrldpc_hip_x.so" if XFLAGS else "libnrldpc_hip.so")  # env override: kernel experiments
OBJDIR = os.path.join(HERE, "build_ab" if AB else "build_allmodes" if ALLMODES else "build_x" if XFLAGS else "build")
SOURCES = ["nrldpc_decode.hip", "nrldpc_encode.hip", "nrldpc_ratematch.hip", "nrldpc_crc.hip", "nrldpc_channel.hip",
           "nrldpc_expand.hip", "nrldpc_capi.hip", "nrldpc_host_quant.cpp"]  # .cpp: host-only C++ (no device pass)
Z64_SOURCE = "nrldpc_decode_z64_inst.hip"
Z64P_SOURCE = "nrldpc_decode_z64p_inst.hip"
# = NRLDPC_Z64_LIST (nrldpc_kernels.h): the sizes where the compile-time-Z kernel beats the run-time-Z one
Z64_BG1 = (60, 64, 104, 112, 120, 128, 144, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384)
Z64_BG2 = (52, 60, 64, 88, 96, 104, 112, 120, 128, 144, 192, 208, 224, 240, 256, 288, 320, 352, 384)
Z64_PAIRS = [(1, z) for z in Z64_BG1] + [(2, z) for z in Z64_BG2]
# = NRLDPC_Z64P_LIST: the packed geometry (a workgroup's row lanes carry several whole codewords)
Z64P_BG1 = (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 44, 48, 52, 56, 72, 80, 88, 96, 176, 352)
Z64P_BG2 = Z64P_BG1[:-4]
Z64P_PAIRS = [(1, z) for z in Z64P_BG1] + [(2, z) for z in Z64P_BG2]
# = NRLDPC_Z64P_NL_LIST: (BG, Z, active layers) with packed builds of their own
Z64P_NL = [(2, 20, 12)]
# = NRLDPC_Z64PR_LIST: (BG, Z, row waves) with pipelined one-thread-per-row builds in the packed geometry
Z64PR = []  # (measured slower than the block-geometry kernels for BG2 88 ... 352: nrldpc_kernels.h)
# = NRLDPC_Z64I_LIST (nrldpc_dispatch_lists.h): (BG, Zr, NCW, mode): the interleaved block geometry -- NCW codewords of the lifting size Zr in one workgroup
# of the block geometry of the virtual size Zr * NCW; mode = what the unit serves (and is compiled for): 1 fixed iteration
# counts, 2 the parity stop with every row active, 4 the parity stop with pruned rows
Z64I = [
    (1, 2, 128, 5), (1, 3, 128, 7), (1, 4, 64, 5), (1, 5, 48, 7), (1, 6, 64, 7), (1, 7, 32, 5), (1, 8, 32, 5), (1, 9, 28, 5), (1, 10, 24, 5), (1, 11, 20, 5),
    (1, 12, 32, 1), (1, 13, 16, 1), (1, 14, 16, 1), (1, 15, 16, 1), (1, 16, 16, 5), (1, 18, 14, 5), (1, 20, 12, 1), (1, 22, 10, 1), (1, 24, 16, 1),
    (1, 26, 8, 1), (1, 28, 8, 1), (1, 30, 8, 1), (1, 32, 8, 5), (1, 36, 7, 7), (1, 40, 6, 5), (1, 44, 5, 7), (1, 48, 8, 1), (1, 52, 4, 5), (1, 56, 4, 5),
    (1, 60, 4, 7), (1, 64, 4, 7), (1, 72, 5, 1), (1, 80, 3, 7), (1, 96, 4, 7), (1, 104, 2, 7), (1, 112, 2, 3), (1, 120, 2, 3), (1, 128, 2, 7), (1, 160, 1, 6),
    (1, 192, 2, 1), (2, 2, 128, 3), (2, 4, 64, 7), (2, 5, 48, 7), (2, 7, 32, 6), (2, 8, 32, 7), (2, 9, 28, 7), (2, 10, 24, 7), (2, 11, 20, 7), (2, 13, 16, 5),
    (2, 14, 16, 5), (2, 15, 16, 5), (2, 16, 16, 5), (2, 18, 14, 5), (2, 20, 12, 5), (2, 22, 10, 5), (2, 26, 8, 5), (2, 28, 8, 5), (2, 30, 8, 5), (2, 32, 8, 5),
    (2, 36, 7, 7), (2, 40, 6, 5), (2, 44, 5, 7), (2, 52, 4, 1), (2, 56, 4, 1), (2, 60, 4, 1), (2, 64, 4, 1), (2, 80, 3, 7), (2, 88, 5, 1), (2, 96, 4, 1),
    (2, 104, 2, 1), (2, 112, 2, 1), (2, 120, 2, 1), (2, 128, 2, 1), (2, 160, 3, 1), (2, 176, 1, 1),
]
# = NRLDPC_Z64_NL_LIST: (BG, Z, active layers) with pipelined kernels of their own
Z64_NL = [(1, 384, 5), (1, 384, 13), (1, 384, 24), (2, 384, 32), (2, 384, 22), (2, 384, 17), (2, 384, 12), (2, 384, 9), (2, 384, 7), (2, 208, 21)]
HEADERS = ["nrldpc_kernels.h", "nrldpc_dispatch_lists.h", "nrldpc_sched.h", "nrldpc_device.h", "nrldpc_decode_z64.h", "nrldpc_decode_z64s.h", "nrldpc_decode_z64p.h", "nrldpc_wave.h", "nrldpc_host_quant.h", "nrldpc_hostpath.h"]
# -mllvm -enable-post-misched=false: LLVM's post-register-allocation machine scheduler off.  The decoder loops are VALU-issue bound and
# hand-ordered (pinned read batches, s_setprio windows, launder() fences); the pre-RA scheduler keeps that order, the post-RA pass
# reshuffles it for latencies the other waves of the CU already hide.  Measured on the MI355X over every lifting size, whole library
# built both ways, order-balanced (profiles/r06_post_ra_scheduler_off.txt): 25 fixed iterations BG1 -3.0 % of the time (geometric mean;
# 44 of 51 sizes by more than 2 %, the headline -2.1 %), BG2 -1.3 %; parity stop at the waterfall BG1 -1.5 %, BG2 -3.2 %; no size more than
# 1.6 % slower.  Seven other scheduling options (max-ilp, max-memory-clause, iterative-ilp, AMDGPU trackers, metric bias, no clustering,
# pre-RA direction) were within noise or worse.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-enable-post-misched=false"] + (["-DNRLDPC_Z64_AB"] if AB else []) + XFLAGS


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


def _deps():
    d = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + [Z64_SOURCE, Z64P_SOURCE]]
    d += [os.path.join(INCLUDE, "nrldpc.h"), os.path.join(INCLUDE, "nr_bg_tables.h"), os.path.abspath(__file__)]
    return [p for p in d if os.path.exists(p)]


def source_id():
    """Hash of every source the library is built from (contents, not mtimes: a snapshot copied to a GPU box keeps
    contents but not times).  Compiled into the library (nrldpc_build_id) and written next to it (<lib>.id)."""
    h = hashlib.sha256()
    for d in sorted(_deps()):
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = ["nrldpc_decode_z64.h", "nrldpc_decode_z64s.h", "nrldpc_decode_z64p.h", "nrldpc_decode_z64_inst.hip",
                  "nrldpc_decode_z64p_inst.hip", "nrldpc_decode.hip",
                  "nrldpc_device.h", "nrldpc_kernels.h", "nrldpc_dispatch_lists.h"]


def kernel_id():
    """Hash of what the DECODER kernels are compiled from (their sources, the base-graph tables, the compiler flags):
    nrldpc_kernel_id() of the library.  rocprofv3 summaries under profiles/ carry it, and bench.py uses a summary's
    instruction counts only when it equals the loaded library's -- a kernel change without a profile refresh cannot
    silently falsify the roofline fractions, while a change elsewhere (C ABI, Python) does not invalidate a profile."""
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, f) for f in KERNEL_SOURCES] + [os.path.join(INCLUDE, "nr_bg_tables.h")]:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


_INC = None


def _includes(path, seen):
    """Files `path` includes with #include "...", transitively (looked up next to it, in csrc/ and in include/)."""
    import re
    try:
        text = open(path, encoding="utf-8", errors="replace").read()
    except OSError:
        return
    for name in re.findall(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', text, re.M):
        for d in (os.path.dirname(path), CSRC, INCLUDE):
            q = os.path.join(d, name)
            if os.path.exists(q):
                q = os.path.abspath(q)
                if q not in seen:
                    seen.add(q)
                    _includes(q, seen)
                break


def _unit_id(src, flags, defs):
    h = hashlib.sha256()
    seen = set()
    _includes(src, seen)
    for f in [src] + sorted(seen):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    h.update(("\0".join([*flags, *defs])).encode())
    return h.hexdigest()[:24]


def _stale():
    """The library is missing, or was built from other sources than the ones in the tree now."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(LIB + ".id") as f:
            return f.read().strip() != source_id()
    except OSError:
        return True


def build_lib(force=False, verbose=False, jobs=None):
    """Compile every HIP source (objects in parallel) and link one shared object, in-tree."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    sid = source_id()
    os.makedirs(OBJDIR, exist_ok=True)
    inc = ["-I" + INCLUDE, "-I" + CSRC]
    units = [(os.path.join(CSRC, f), os.path.join(OBJDIR, os.path.splitext(f)[0] + ".o"),
              ['-DNRLDPC_BUILD_ID="%s"' % sid, '-DNRLDPC_KERNEL_ID="%s"' % kernel_id()] if f == "nrldpc_capi.hip" else
             ["-DNRLDPC_Z64I_FORCE_MODE=7"] if (ALLMODES and f == "nrldpc_decode.hip") else [])
             for f in SOURCES]
    units += [(os.path.join(CSRC, Z64_SOURCE), os.path.join(OBJDIR, "z64_%d_%d.o" % (bg, z)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % z]) for bg, z in Z64_PAIRS]
    units += [(os.path.join(CSRC, Z64P_SOURCE), os.path.join(OBJDIR, "z64p_%d_%d.o" % (bg, z)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % z]) for bg, z in Z64P_PAIRS]
    units += [(os.path.join(CSRC, Z64P_SOURCE), os.path.join(OBJDIR, "z64p_%d_%d_nl%d.o" % (bg, z, nl)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % z, "-DNRLDPC_Z64_NL=%d" % nl]) for bg, z, nl in Z64P_NL]
    units += [(os.path.join(CSRC, Z64P_SOURCE), os.path.join(OBJDIR, "z64pr_%d_%d.o" % (bg, z)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % z, "-DNRLDPC_Z64P_ROW=1", "-DNRLDPC_Z64P_RW=%d" % rw]) for bg, z, rw in Z64PR]
    units += [(os.path.join(CSRC, Z64P_SOURCE), os.path.join(OBJDIR, "z64i_%d_%d.o" % (bg, z)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % (z * ncw), "-DNRLDPC_Z64_ILV=%d" % ncw, "-DNRLDPC_Z64I_ZR=%d" % z,
                "-DNRLDPC_Z64S_DUAL=0", "-DNRLDPC_Z64I_MODE=%d" % (7 if ALLMODES else mode)]) for bg, z, ncw, mode in Z64I]
    units += [(os.path.join(CSRC, Z64_SOURCE), os.path.join(OBJDIR, "z64_%d_%d_nl%d.o" % (bg, z, nl)),
               ["-DNRLDPC_Z64_BG=%d" % bg, "-DNRLDPC_Z64_Z=%d" % z, "-DNRLDPC_Z64_NL=%d" % nl]) for bg, z, nl in Z64_NL]
    # every decoder unit gets a name space of its own for the -D-dependent templates (NRLDPC_UNIT, nrldpc_decode_z64.h)
    units = [(src, obj, defs + (["-DNRLDPC_UNIT=u_" + os.path.splitext(os.path.basename(obj))[0]] if os.path.basename(src) in (Z64_SOURCE, Z64P_SOURCE) else []))
             for src, obj, defs in units]
    # An object is reused only when it was compiled from exactly these inputs: contents of its source and of every header
    # it includes (transitively, by scanning #include "..." lines: a change to the C ABI does not recompile 130 decoder
    # units), the flags and the -D list (sidecar <obj>.id) -- never by modification time, which a snapshot copy, rsync -t or
    # tar may set to anything.  nrldpc_capi carries the build id of the whole tree in its -D list, so it is keyed by that too.
    def unit_id(src, flags, defs):
        return _unit_id(src, flags, defs)

    def compile_one(u):
        src, obj, defs = u
        flags = [f for f in FLAGS if not f.startswith("--offload-arch") and f not in ("-mllvm", "-enable-post-misched=false")] if src.endswith(".cpp") else FLAGS  # host-only C++: no device pass, the compiler's own schedule
        uid = unit_id(src, flags, defs)
        try:
            with open(obj + ".id") as f:
                fresh = os.path.exists(obj) and f.read().strip() == uid
        except OSError:
            fresh = False
        if fresh and not force:
            return obj
        cmd = [hipcc, *flags, *inc, *defs, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(obj + ".id"):
            os.remove(obj + ".id")
        subprocess.check_call(cmd)
        with open(obj + ".id", "w") as f:
            f.write(uid + "\n")
        return obj

    jobs = jobs or min(len(units), max(1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(LIB + ".id", "w") as f:
        f.write(sid + "\n")
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
