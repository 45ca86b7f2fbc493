"""Stress config C5 (BASELINE.json configs[4] / SURVEY 8d): n=1e6, m=8000, d=1024, p=6 with BOTH engines: fp64 end to end
(X, Y 16 GB resident, feature matrix built in 48-GB passes) and the fp32 engine BASELINE names for this configuration
(nk_set_compute_dtype: kernel blocks and Gram contractions in fp32, Gram accumulators and everything m x m in fp64).
Reports stage times of the second fit of each and the fp32 engine's predictions / 20-step forecast against the fp64
engine's (no CPU reference exists at this size).   python3 tools/stress_c5.py [n m d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nys_koop_lqr_amd as nk
n, m, d = [int(v) for v in (sys.argv[1:4] + ["1000000", "8000", "1024"][len(sys.argv) - 1:])]
p = 6
t0 = time.perf_counter()
rng = np.random.default_rng(1234)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
S = torch.randn((n, d), dtype=torch.float64, device=dev, generator=gen)
U = torch.randn((n, p), dtype=torch.float64, device=dev, generator=gen)
Wt = torch.randn((d, d), dtype=torch.float64, device=dev, generator=gen) * (0.9 / np.sqrt(d))
Bt = torch.randn((p, d), dtype=torch.float64, device=dev, generator=gen) * 0.1
Y = torch.empty((n, d), dtype=torch.float64, device=dev)
for r in range(0, n, 100000):  # chunked so that torch's GEMM workspace stays small
    Y[r:r + 100000] = torch.tanh(S[r:r + 100000] @ Wt) + U[r:r + 100000] @ Bt
X = torch.cat([S, U], dim=1).contiguous()
del S
torch.cuda.synchronize()
print(f"data on device in {time.perf_counter() - t0:.1f} s (synthetic, generated with torch on the GPU: plumbing only)", flush=True)
np.random.seed(0)
idx = np.random.choice(np.arange(n), size=m, replace=False)
Z = Y[torch.from_numpy(idx).to(dev)].cpu().numpy()
ls = 20.0 * np.sqrt(d / 384.0)  # same kernel width per dimension as C4
mp = m + p
flop = (mp * (mp + 1) + 2.0 * m * mp + m * (m + 1) + 2.0 * d * m) * n
q = torch.from_numpy(np.random.default_rng(0).choice(n, 2000, replace=False)).to(dev)
Xq, Yq = X[q].cpu().numpy(), Y[q].cpu().numpy()
Useq = np.random.default_rng(1).standard_normal((p, 20))
x0 = X[7, :d].cpu().numpy()
res = {}
for dtype in ("f64", "f32"):
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=1e-6, m=m)
    reg.compute_dtype = dtype
    reg.nystrom_centers_output = Z.T
    for rep in range(2):
        t1 = time.perf_counter(); reg.fit(X, Y); t2 = time.perf_counter()
        st = reg.fit_stats_
        print(f"[{dtype}] fit {rep}: wall {t2 - t1:.2f} s | device {st['ms_total'] / 1e3:.2f} s: kmat(first pass) {st['ms_kmat'] / 1e3:.2f} gram+later passes {st['ms_gram'] / 1e3:.2f} "
              f"sqrt {st['ms_sqrt'] / 1e3:.2f} (iters {st['sqrt_iters']}, res {st['sqrt_residual']:.1e}) | gram launches {st['gram_kernel_launches']} "
              f"avg {st['ms_gram_kernel_avg']:.1f} ms | ranks {st['rank_inner']}/{st['rank_inner_rec']}", flush=True)
    peak = 78.6 if dtype == "f64" else 157.3
    tf = flop / (st['ms_gram_kernel_avg'] * st['gram_kernel_launches'] * 1e-3) / 1e12
    print(f"[{dtype}] Gram algorithmic flop {flop:.3e}; at the measured kernel time: {tf:.1f} TFLOP/s = {tf / peak:.2f} of the {dtype} matrix peak ({peak} TF)")
    pred = reg.predict(Xq)
    sim = reg.rollout(x0, Useq)
    print(f"[{dtype}] one-step prediction on 2000 training rows: relF vs targets %.3e; W == C [A B]: %.2e" %
          (np.linalg.norm(pred - Yq) / np.linalg.norm(Yq), np.linalg.norm(reg.weights - reg.C @ np.hstack([reg.A, reg.B])) / np.linalg.norm(reg.weights)), flush=True)
    res[dtype] = (pred, sim, np.array(reg.B))
    del reg
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("fp32 engine vs fp64 engine: predictions %.3e, 20-step forecast %.3e, B %.3e" %
      (rel(res["f32"][0], res["f64"][0]), rel(res["f32"][1], res["f64"][1]), rel(res["f32"][2], res["f64"][2])))
