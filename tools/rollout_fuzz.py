"""Random shapes through nk_linear_rollout / closed loops against NumPy loops: both single-launch recursions (m <= 128 in one
workgroup, m <= 2048 over several) and the per-step paths, single and batched.  python3 tools/rollout_fuzz.py [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(11)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ms = [1, 2, 7, 16, 33, 64, 100, 127, 128, 129, 130, 160, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 2300]
worst = 0.0
for it in range(count):
    m = int(rng.choice(ms)); p = int(rng.choice([0, 1, 2, 6, 31, 64])); d = int(rng.choice([1, 3, 9, 192]))
    T = int(rng.choice([1, 2, 3, 4, 17, 60])); batch = int(rng.choice([1, 1, 2, 5, 16, 17, 70]))
    if m > 1024 and batch > 17: batch = 17
    A = rng.standard_normal((m, m)) * (0.9 / np.sqrt(m)); B = rng.standard_normal((m, p)); Cm = rng.standard_normal((d, m))
    z0 = rng.standard_normal((batch, m)); U = rng.standard_normal((batch, T, p))
    out, outz = nk.linear_rollout(A, B, Cm, z0, U, return_lifted=True)
    z = z0.copy(); ref = np.empty((batch, T, d)); refz = np.empty((batch, T, m))
    for t in range(T):
        refz[:, t, :] = z; ref[:, t, :] = z @ Cm.T
        z = z @ A.T + (U[:, t, :] @ B.T if p else 0.0)
    e = max(np.linalg.norm(out - ref) / max(np.linalg.norm(ref), 1e-300), np.linalg.norm(outz - refz) / np.linalg.norm(refz))
    worst = max(worst, e)
    flag = "" if e < 1e-11 else "   <-- LARGE"
    print(f"m={m:5d} p={p:2d} d={d:3d} T={T:3d} batch={batch:3d}: err {e:.1e}{flag}", flush=True)
print("worst", worst)
sys.exit(0 if worst < 1e-11 else 1)
