#!/bin/bash
# Spread of the host-pointer path from process to process: tools/bench_host_path.py --big-only, N processes per setting.
cd ${GRAFT_REPO_ROOT:-/root/repo}
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); print("  %-8s min %.2f med %.2f max %.2f" % (r["llr_dtype"], r["ms_min"], r["ms_median"], r["ms_max"]))'
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)  nproc: $(nproc)  loadavg: $(cat /proc/loadavg)"
for spin in ${SPINS:-200 0 50}; do for r in 1 2 3; do echo "spin ${spin} us, process $r"; NRLDPC_HOST_SPIN_US=$spin python tools/bench_host_path.py --big-only 2>/dev/null | python -c "$fmt"; done; done
echo "native format"; for r in 1 2; do NRLDPC_HOST_I8=0 python tools/bench_host_path.py --big-only 2>/dev/null | python -c "$fmt"; done
