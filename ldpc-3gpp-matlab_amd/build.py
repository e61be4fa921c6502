"""Build recipe for libnrldpc_hip.so (gfx950 only; hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.environ.get("NRLDPC_LIB") or os.path.join(HERE, "libnrldpc_hip.so")  # env override: kernel experiments
SOURCES = ["nrldpc_decode.hip", "nrldpc_decode_z64.hip", "nrldpc_encode.hip", "nrldpc_ratematch.hip",
           "nrldpc_crc.hip", "nrldpc_capi.hip"]
HEADERS = ["nrldpc_kernels.h", "nrldpc_sched.h", "nrldpc_device.h"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, f))]
    deps += [os.path.join(INCLUDE, "nrldpc.h"), os.path.join(INCLUDE, "nr_bg_tables.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every HIP source into one shared object, in-tree."""
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + INCLUDE, "-I" + CSRC, *srcs, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
