import sys, time, importlib, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
A = importlib.import_module("ldpc-3gpp-matlab_amd")
import oracle as O
rng = np.random.default_rng(3)
def run(bg, Z, B, EsN0, it, nl=0, et=False, dt=np.float16, app=True):
    kb = O.BG_DIMS[bg][2]
    info = rng.integers(0, 2, (B, kb*Z), dtype=np.uint8)
    cw = O.encode(bg, Z, info)
    c = A.Codec(bg, Z, max_iter=it, n_layers=nl, early_term=et, llr_dtype=dt)
    cwg = c.encode(info)
    enc_ok = bool((cwg == cw).all())
    mu = 2*10**(EsN0/10)
    llr = ((1-2.0*cw)*mu + np.sqrt(2*mu)*rng.standard_normal(cw.shape)).astype(dt)
    llr[:, :2*Z] = 0
    t=time.time(); hg, ig, ag = c.decode(llr, True, True); tg=time.time()-t
    ho, io, ao = O.decode_nmsq(bg, Z, llr.astype(np.float64), it, n_layers=nl, early_term=et, want_app=True)
    print(f"bg{bg} Z{Z} B{B} it{it} nl{nl} et{et} {np.dtype(dt).name}: enc_ok={enc_ok} hard_eq={(hg==ho).all()} iters_eq={(ig==io).all()} app_eq={(ag==ao).all()} maxappdiff={np.abs(ag-ao).max()} ber={(hg!=info).mean():.4f} t={tg:.3f}", flush=True)
    c.close()
run(1, 384, 4, -1.0, 5)
run(1, 384, 4, -1.0, 25, et=True)
run(2, 384, 4, -1.0, 10, dt=np.float32)
run(2, 20, 7, 1.0, 10, et=True)
run(1, 2, 300, 3.0, 10, et=True)
run(1, 208, 5, 0.0, 8, nl=20)
run(2, 96, 9, 2.0, 12, nl=8, et=True, dt=np.float32)
