"""Multi-workgroup recursion (m > 128): which shapes work, how long a step takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(0)
def tm(f, reps=10):
    f(); f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]
cases = [(256, 2, 0), (500, 6, 0), (500, 6, 6), (520, 6, 6), (1000, 8, 3), (2000, 8, 6)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (m, d, p) in cases:
    A = rng.standard_normal((m, m)) * (0.9 / np.sqrt(m)); B = rng.standard_normal((m, p)); Cm = rng.standard_normal((d, m))
    for T in (3, 100):
        for batch in (1, 4, 16, 64):
            z0 = rng.standard_normal((batch, m)); U = rng.standard_normal((batch, T, p))
            try:
                out = nk.linear_rollout(A, B, Cm, z0, U) if batch > 1 else nk.linear_rollout(A, B, Cm, z0[0].reshape(-1, 1), U[0].T)
            except Exception as e:
                print(f"m={m} p={p} T={T} batch={batch}: FAILED {e}", flush=True)
                continue
            z = z0.copy(); ref = np.empty((batch, T, d))
            for t in range(T):
                ref[:, t, :] = z @ Cm.T
                z = z @ A.T + U[:, t, :] @ B.T
            got = out if batch > 1 else out.T[None]
            err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            ms, lo, hi = tm(lambda: nk.linear_rollout(A, B, Cm, z0, U) if batch > 1 else nk.linear_rollout(A, B, Cm, z0[0].reshape(-1, 1), U[0].T), 15)
            print(f"m={m} p={p} T={T} batch={batch}: err {err:.1e}  median {ms:.3f} ms (min {lo:.3f}, max {hi:.3f}) per call ({ms / max(T - 1, 1) * 1e3:.2f} us per step)", flush=True)
