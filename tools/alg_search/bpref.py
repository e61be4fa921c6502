import sys, json
import search as S
case = sys.argv[1]; nblk = int(sys.argv[2])
for it in (S.CASES[case][5], 50):
    print(case, "BP", it, " ".join("%.1f:%.4f(%.1f)" % (s, *[S.bp_ref(case, s, nblk, it)[k] for k in ("bler", "it")]) for s in S.CASES[case][6]), flush=True)
