#!/usr/bin/env python3
"""Static instruction mix of the headline kernel's iteration loop, priced with the measured gfx950 issue rates.

Compiles the (BG, Z) instance device-only (no GPU needed), disassembles the fixed-iteration kernel, takes the
iteration loop (the longest backward branch), counts instructions per opcode and multiplies each VALU opcode by the
issue interval measured on the MI355X (profiles/r02_ubench_valu_rates.txt, tools/ubench/valu_rate*.hip).  The sum is
the time the loop needs if the VALU pipes never idle: the cycle-weighted VALU roofline of this kernel, to be set
against the measured kernel time (bench.py / profiles/r02_bench_kernel_stats.csv) and against SQ_ACTIVE_INST_VALU
(profiles/r02_bench_pmc_summary.json).  The per-wave mirror-coherence dispatch inside the loop holds only LDS writes,
scalar compares and branches, so the static VALU count of the loop is the count every wave executes.

--form row: the one-thread-per-row kernel (one loop).  --form split: the two-threads-per-row kernel
(nrldpc_decode_z64s.h): one loop per half; a row's work per iteration is the SUM of both loops.

    python tools/isa_mix.py [--bg 1 --z 384] [--form split] [--kernel-ms 3.3] [--json profiles/r03_headline_isa_mix.json] > profiles/r03_headline_isa_mix.txt
"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ldpc-3gpp-matlab_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def rates():
    """mnemonic -> issue interval in ns per wave64 instruction per SIMD (the table quotes cycles at a nominal 2.4 GHz)."""
    tab = {}
    path = os.path.join(ROOT, "profiles", "r03_ubench_valu_rates.txt")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r02_ubench_valu_rates.txt")
    for line in open(path):
        m = re.match(r"(v_[a-z0-9_]+)[^>]*-> ([0-9.]+) cycles/inst", line)
        if m and m.group(1) not in tab and not (m.group(1).startswith("v_cndmask") and float(m.group(2)) > 10):  # (r02's v_cndmask lines measured a vcc dependency chain)
            tab[m.group(1)] = float(m.group(2)) / 2.4
    return tab


def price(op, tab):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa"):
        return tab.get(op, tab.get("v_cvt_f32_i32_sdwa")), "sdwa"
    if base in tab:
        return tab[base], "measured"
    if base == "v_cndmask_b32":
        return tab["v_mov_b32"], "as v_mov_b32"     # no independent-chain measurement in the table: priced as v_mov_b32
    if base in ("v_max_f32", "v_min_f32"):
        return tab["v_min_f32"], "measured"
    return tab["v_add_f32"], "assumed full rate"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bg", type=int, default=1)
    ap.add_argument("--z", type=int, default=384)
    ap.add_argument("--kernel-ms", type=float, default=None, help="measured fixed-25 kernel time of the batch below; default: "
                    "roofline.kernel_ms of --bench-line (no number is assumed: without either, no fraction is printed)")
    ap.add_argument("--bench-line", default=os.path.join(ROOT, "gpurun_out", "bench_line_noprofile.json"),
                    help="a bench.py line of the same build and session to take the kernel time from")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--launch-invariant-ms", type=float, default=0.06)
    ap.add_argument("--form", default="split", choices=("row", "split"))
    ap.add_argument("--json", default=None, help="also write the totals (with nrldpc_kernel_id) for bench.py")
    a = ap.parse_args()
    kms_src = "--kernel-ms"
    if a.kernel_ms is None and os.path.exists(a.bench_line):
        try:
            import json as _json
            a.kernel_ms = float(_json.load(open(a.bench_line))["roofline"]["kernel_ms"])
            kms_src = "roofline.kernel_ms of " + os.path.relpath(a.bench_line, ROOT)
        except Exception:
            a.kernel_ms = None
    tab = rates()
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "k.co")
        import importlib
        sys.path.insert(0, ROOT)
        flags = [f for f in importlib.import_module("ldpc-3gpp-matlab_amd.build").FLAGS if f != "-fPIC"]  # the library's own flags (scheduling options included)
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-DNRLDPC_Z64_BG=%d" % a.bg, "-DNRLDPC_Z64_Z=%d" % a.z, "--cuda-device-only",
                               "--no-gpu-bundle-output", "-c", os.path.join(CSRC, "nrldpc_decode_z64_inst.hip"), "-o", co])
        dis = subprocess.check_output([OBJDUMP, "-d", co], text=True)
    # the fixed-iteration build: row form <BG, Z, NCWG, FULL=1, PLAIN=1, ETP=0, NL>; split form <BG, Z, ETP=0, NL>
    # (... with every row active: NL = 46 / 42 -- the unit also holds the builds with a run-time layer count, NL = 0)
    rows_all = {1: 46, 2: 42}[a.bg]
    pat = (r"nrldpc_decode_z64_kernelILi%dELi%dELi\d+ELb1ELb1ELb0ELi%dE" if a.form == "row" else r"nrldpc_decode_z64s_kernelILi%dELi%dELb0ELi%dE") % (a.bg, a.z, rows_all)
    cur, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if cur and re.search(pat, cur):
            m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m:
                body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    assert body, "kernel not found"
    spans = []  # backward branches: the longest closes the iteration loop (split form: one loop per half)
    for ad, op, args in body:
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            if off >= 32768:
                spans.append((ad + 4 + (off - 65536) * 4, ad))
    # the iteration loop(s): the innermost backward branches that enclose exactly one barrier per barrier group (BG1 32,
    # BG2 28 with every row active) -- one loop in the row form, one per half in the split form
    ng = {1: 32, 2: 28}[a.bg]
    ng1 = {1: 46, 2: 42}[a.bg]  # the split form's one-layer groups (z64s_single)
    bars = [b[0] for b in body if b[1] == "s_barrier"]
    want = 2 if a.form == "split" else 1
    keep = []
    for nb in ((ng,) if a.form == "row" else (ng, ng + 4, ng1, ng1 + 4)):  # + 4: the split form's dual rows 0..3 have a second barrier each (Own::dual)
        cand = sorted((sp for sp in spans if sum(1 for x in bars if sp[0] <= x <= sp[1]) == nb), key=lambda t: t[1] - t[0])
        keep = []
        for sp in cand:
            if all(sp[1] < k[0] or sp[0] > k[1] for k in keep):
                keep.append(sp)
        if len(keep) == want:
            break
    assert len(keep) == want, "expected %d iteration loop(s) with %d (+4) barriers, found %d" % (want, ng, len(keep))
    loop = [b for b in body if any(lo <= b[0] <= hi for lo, hi in keep)]
    lo, hi = min(k[0] for k in keep), max(k[1] for k in keep)
    loop_bytes = sum(k[1] - k[0] for k in keep)
    cnt = collections.Counter(op for _, op, _ in loop)
    cls = collections.Counter()
    for op, n in cnt.items():
        c = "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "barrier" if op == "s_barrier" else \
            "waitcnt/nop" if op in ("s_waitcnt", "s_nop") else "SALU/branch" if op.startswith("s_") else "VMEM"
        cls[c] += n
    print("# tools/isa_mix.py: iteration loop%s of the %s kernel, BG=%d Z=%d, fixed iteration count (device-only compile of the tree's sources)" % (
        "s (one per half, summed)" if a.form == "split" else "", "two-threads-per-row (split)" if a.form == "split" else "one-thread-per-row", a.bg, a.z))
    print("# loop = code between a long backward branch and its target: %d instructions, %.1f KB in total" % (len(loop), loop_bytes / 1024.0))
    print("instruction classes per iteration, summed over the waves that serve one block of 64 rows:", dict(cls))
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
    print("VALU-bound time of the timed launch: %.1f ns x %.1f row blocks per SIMD (2 codewords per CU) x %d iterations x %.1f rounds = %.3f ms" % (
        total, wps, a.iters, rounds, valu_ms))
    if a.kernel_ms:
        print("measured kernel time %.3f ms (%s; of which %.2f ms launch-invariant: workgroup start, LLR ingest, write-back)" % (
            a.kernel_ms, kms_src, a.launch_invariant_ms))
        print("cycle-weighted VALU roofline fraction: %.3f of the kernel, %.3f of its iteration part" % (
            valu_ms / a.kernel_ms, valu_ms / (a.kernel_ms - a.launch_invariant_ms)))
    else:
        print("(no measured kernel time given: bench.py divides by the time it measures itself -- roofline.cycle_weighted)")
    print("(the 2-cycle-per-op VALU-issue fraction of the bench line prices every op at the full rate)")
    if a.json:
        import importlib, json
        sys.path.insert(0, ROOT)
        kid = importlib.import_module("ldpc-3gpp-matlab_amd.build").kernel_id()
        json.dump({"nrldpc_kernel_id": kid, "bg": a.bg, "Z": a.z, "form": a.form, "valu_instructions_per_iteration_all_waves_of_a_row": cls["VALU"],
                   "valu_ns_per_iteration_all_waves_of_a_row": total, "lds_instructions": cls["LDS"], "barriers": cls["barrier"],
                   "loop_bytes": loop_bytes, "source": "tools/isa_mix.py (static disassembly x measured issue intervals)"},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
