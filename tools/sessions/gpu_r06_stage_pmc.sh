#!/bin/bash
# SQ / memory counters of the stage kernels (tools/bench_chain.py): what each is bound by
mkdir -p gpurun_out/r06z; cd /tmp; export TMPDIR=/tmp
R=/root/repo
pmc() { local name=$1; shift; OUT_SUFFIX=_pmc rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/z6_$name -o $name -- python $R/tools/bench_chain.py > /tmp/z6_$name.log 2>&1; }
pmc sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY
pmc sqB SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python - <<'PY' > $R/gpurun_out/r06z/stage_pmc.txt 2>&1
import csv, glob, collections
for name in ("sqA", "sqB", "grbm", "fetch", "write"):
    f = glob.glob("/tmp/z6_%s/**/*counter_collection.csv" % name, recursive=True)
    if not f: print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "nrldpc" not in k or "decode" in k: continue
        key = k[k.find("nrldpc_"):][:48] + " grid%s" % r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (key, r["Dispatch_Id"]) not in seen: seen.add((key, r["Dispatch_Id"])); n[key] += 1
    for key in sorted(acc):
        print(name, key, "n", n[key], {c: round(v / n[key]) for c, v in sorted(acc[key].items())})
PY
grep -c . $R/gpurun_out/r06z/stage_pmc.txt; grep "rate_recover_fast\|rate_match" $R/gpurun_out/r06z/stage_pmc.txt | cut -c1-400 | head -30
