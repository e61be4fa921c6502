#!/bin/bash
# counters of the headline code's kernel under the parity stop (the reference's only mode) next to the fixed-iteration launch
mkdir -p gpurun_out/r06ab; cd /tmp; export TMPDIR=/tmp
R=/root/repo
B="python $R/bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-e2e"
pmc() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/ab6_$name -o $name -- $B > /tmp/ab6_$name.log 2>&1; }
pmc sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY
pmc sqB SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python - <<'PY' > $R/gpurun_out/r06ab/headline_stop_pmc.txt 2>&1
import csv, glob, collections
for name in ("sqA", "sqB", "grbm", "fetch", "write"):
    f = glob.glob("/tmp/ab6_%s/**/*counter_collection.csv" % name, recursive=True)
    if not f: print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "nrldpc_decode_z64s_kernel" not in k: continue
        key = k[k.find("nrldpc_decode"):][:70]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (key, r["Dispatch_Id"]) not in seen: seen.add((key, r["Dispatch_Id"])); n[key] += 1
    for key in sorted(acc):
        print(name, key, "n", n[key], {c: round(v / n[key]) for c, v in sorted(acc[key].items())})
PY
cat $R/gpurun_out/r06ab/headline_stop_pmc.txt | cut -c1-400
