#!/bin/bash
# Host-pointer path (nrldpc_decode, batch 4096 headline codewords): int8 on the wire (default) against the native
# format (NRLDPC_HOST_I8=0), copy-thread counts; min / median / max over 12 calls each.
cd ${GRAFT_REPO_ROOT:-/root/repo}
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print("  %-8s min %.2f med %.2f max %.2f ms  %.2f Gbit/s (first call %.1f ms)" % (r["llr_dtype"], r["ms_min"], r["ms_median"], r["ms_max"], r["info_Gbit_s_median"], r["ms_first_call"]))'
for th in 8 16 24 32; do
  echo "int8 on the wire, $th copy threads"; NRLDPC_HOST_THREADS=$th python tools/bench_host_path.py --big-only 2>/dev/null | python -c "$fmt"
done
echo "native format (NRLDPC_HOST_I8=0), 16 copy threads"; NRLDPC_HOST_I8=0 NRLDPC_HOST_THREADS=16 python tools/bench_host_path.py --big-only 2>/dev/null | python -c "$fmt"
