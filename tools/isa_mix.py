#!/usr/bin/env python3
"""Static instruction mix of the headline kernel's iteration loop, priced with the measured gfx950 issue rates.

Compiles the (BG, Z) instance device-only (no GPU needed), disassembles the fixed-iteration kernel, takes the
iteration loop (the longest backward branch), counts instructions per opcode and multiplies each VALU opcode by the
issue interval measured on the MI355X (profiles/r02_ubench_valu_rates.txt, tools/ubench/valu_rate*.hip).  The sum is
the time the loop needs if the VALU pipes never idle: the cycle-weighted VALU roofline of this kernel, to be set
against the measured kernel time (bench.py / profiles/r02_bench_kernel_stats.csv) and against SQ_ACTIVE_INST_VALU
(profiles/r02_bench_pmc_summary.json).  The per-wave mirror-coherence dispatch inside the loop holds only LDS writes,
scalar compares and branches, so the static VALU count of the loop is the count every wave executes.

    python tools/isa_mix.py [--bg 1 --z 384] [--kernel-ms 4.03] > profiles/r02_headline_isa_mix.txt
"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ldpc-3gpp-matlab_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def rates():
    """mnemonic -> issue interval in ns per wave64 instruction per SIMD (the table quotes cycles at a nominal 2.4 GHz)."""
    tab = {}
    for line in open(os.path.join(ROOT, "profiles", "r02_ubench_valu_rates.txt")):
        m = re.match(r"(v_[a-z0-9_]+)[^>]*-> ([0-9.]+) cycles/inst", line)
        if m and m.group(1) not in tab and "cndmask" not in m.group(1):  # (the v_cndmask lines measure a vcc dependency chain)
            tab[m.group(1)] = float(m.group(2)) / 2.4
    return tab


def price(op, tab):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa"):
        return tab.get(op, tab.get("v_cvt_f32_i32_sdwa")), "sdwa"
    if base in tab:
        return tab[base], "measured"
    if base == "v_cndmask_b32":
        return tab["v_mov_b32"], "as v_mov_b32"     # full rate; its ubench line is a dependency chain, not an issue rate
    if base in ("v_max_f32", "v_min_f32"):
        return tab["v_min_f32"], "measured"
    return tab["v_add_f32"], "assumed full rate"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bg", type=int, default=1)
    ap.add_argument("--z", type=int, default=384)
    ap.add_argument("--kernel-ms", type=float, default=3.67, help="measured fixed-25 kernel time of the batch below")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--launch-invariant-ms", type=float, default=0.06)
    a = ap.parse_args()
    tab = rates()
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "k.co")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-DNRLDPC_Z64_BG=%d" % a.bg, "-DNRLDPC_Z64_Z=%d" % a.z, "--cuda-device-only",
                               "--no-gpu-bundle-output", "-c", os.path.join(CSRC, "nrldpc_decode_z64_inst.hip"), "-o", co])
        dis = subprocess.check_output([OBJDUMP, "-d", co], text=True)
    # the fixed-iteration build: template arguments <BG, Z, NCWG, FULL=1, PLAIN=1, ETP=0, NL>
    cur, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if cur and re.search(r"nrldpc_decode_z64_kernelILi%dELi%dELi\d+ELb1ELb1ELb0E" % (a.bg, a.z), cur):
            m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m:
                body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    assert body, "kernel not found"
    addr = [b[0] for b in body]
    best = None  # the backward branch with the longest span closes the iteration loop
    for ad, op, args in body:
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            if off >= 32768:
                tgt = ad + 4 + (off - 65536) * 4
                if best is None or ad - tgt > best[1] - best[0]:
                    best = (tgt, ad)
    lo, hi = best
    loop = [b for b in body if lo <= b[0] <= hi]
    cnt = collections.Counter(op for _, op, _ in loop)
    cls = collections.Counter()
    for op, n in cnt.items():
        c = "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "barrier" if op == "s_barrier" else \
            "waitcnt/nop" if op in ("s_waitcnt", "s_nop") else "SALU/branch" if op.startswith("s_") else "VMEM"
        cls[c] += n
    print("# tools/isa_mix.py: iteration loop of nrldpc_decode_z64_kernel<BG=%d, Z=%d, FULL, PLAIN> (device-only compile of the tree's sources)" % (a.bg, a.z))
    print("# loop = code between the longest backward branch and its target: %d instructions, %.1f KB" % (len(loop), (hi - lo) / 1024.0))
    print("instruction classes per iteration and wave:", dict(cls))
    edges = {1: 316, 2: 197}[a.bg]
    rows = {1: 46, 2: 42}[a.bg]
    print("VALU per edge and iteration: %.2f (%d edges, %d check rows per thread)" % (cls["VALU"] / edges, edges, rows))
    print()
    print("%-28s %6s %9s %10s  %s" % ("VALU opcode", "count", "ns/inst", "ns/iter", "rate source"))
    total = 0.0
    rows_out = []
    for op, n in cnt.items():
        if not op.startswith("v_"):
            continue
        r, src = price(op, tab)
        rows_out.append((n * r, op, n, r, src))
        total += n * r
    for t, op, n, r, src in sorted(rows_out, reverse=True):
        print("%-28s %6d %9.3f %10.1f  %s" % (op, n, r, t, src))
    print("%-28s %6d %9s %10.1f" % ("sum", cls["VALU"], "", total))
    nwv = a.z // 64
    # waves per SIMD that share the VALU: one workgroup of 2 codewords (2*nwv waves) per CU over 4 SIMDs
    wps = 2 * nwv / 4.0
    cus = 256
    rounds = a.batch / 2.0 / cus
    valu_ms = total * wps * a.iters * rounds * 1e-6
    print()
    print("VALU-bound time of the timed launch: %.1f ns x %.1f waves per SIMD x %d iterations x %.1f workgroup rounds = %.3f ms" % (
        total, wps, a.iters, rounds, valu_ms))
    print("measured kernel time %.3f ms (of which %.2f ms launch-invariant: workgroup start, LLR ingest, write-back)" % (a.kernel_ms, a.launch_invariant_ms))
    print("cycle-weighted VALU roofline fraction: %.3f of the kernel, %.3f of its iteration part" % (
        valu_ms / a.kernel_ms, valu_ms / (a.kernel_ms - a.launch_invariant_ms)))
    print("(compare SQ_ACTIVE_INST_VALU / busy cycles = valu_pipe_busy in profiles/r02_bench_pmc_summary.json; the 2-cycle-per-op")
    print(" VALU-issue fraction of the bench line prices every op at the full rate and reads 0.40 for the same launch)")


if __name__ == "__main__":
    main()
