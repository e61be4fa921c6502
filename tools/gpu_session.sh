cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/s11_et.txt 2>&1
import importlib, os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.getcwd())
# realistic early-termination timing: QPSK/AWGN codewords at a waterfall point (bench_one uses random LLRs that never converge)
code = r'''
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
bg, Z, nl, E, snr, B = [float(x) if i == 4 else int(x) for i, x in enumerate(sys.argv[1:7])]
rows, cols, kb = {1: (46, 68, 22), 2: (42, 52, 10)}[bg]
c = pkg.Codec(bg, Z, max_iter=25, n_layers=nl, early_term=True, llr_dtype=np.float16)
g = torch.Generator(device="cuda"); g.manual_seed(5)
info = torch.randint(0, 2, (B, kb * Z), generator=g, device="cuda", dtype=torch.uint8)
cw = torch.empty((B, cols * Z), device="cuda", dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream
c.encode_dev(info.data_ptr(), B, cw.data_ptr(), s)
mu = 2 * 10 ** (snr / 10)
llr = (1 - 2 * cw.float()) * mu + (2 * mu) ** 0.5 * torch.randn(cw.shape, generator=g, device="cuda")
llr[:, :2 * Z] = 0; llr[:, 2 * Z + E:] = 0
llr = llr.half().contiguous()
hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8); it = torch.empty(B, device="cuda", dtype=torch.int32)
c.set_timing(True); ms = []
for i in range(7):
    c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), it.data_ptr(), None, s); ms.append(c.last_kernel_ms())
print("%-28s BG%d Z=%d nl=%d %.1f dB: %.3f ms  %.2f Gbit/s  mean it %.2f bler %.4f" % (os.path.basename(os.environ.get("NRLDPC_LIB", "default")), bg, Z, nl, snr, min(ms[1:]), B * kb * Z / min(ms[1:]) / 1e6, it.float().mean().item(), (hard != info).any(1).float().mean().item()))
'''
open("/tmp/et_one.py", "w").write(code)
cases = [((1, 384, 0, 25344, -0.5, 4096), "lib_1_384_1_3.so"), ((1, 384, 0, 25344, -1.2, 4096), "lib_1_384_1_3.so"),
         ((2, 384, 0, 19120, -3.0, 4096), "lib_2_384_1_6.so"), ((1, 256, 0, 16896, -0.5, 4096), "lib_1_256_1_4.so"),
         ((2, 384, 22, 11472, -0.5, 4096), "lib_2_384_nl22_1_6.so")]
for args, lib in cases:
    for l in (None, lib):
        env = dict(os.environ)
        if l: env["NRLDPC_LIB"] = os.path.join(os.getcwd(), "exp_libs", l)
        subprocess.run([sys.executable, "/tmp/et_one.py"] + [str(a) for a in args], env=env)
PY
grep -v amdgpu gpurun_out/s11_et.txt
