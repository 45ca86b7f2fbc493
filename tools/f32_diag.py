#!/usr/bin/env python3
"""fp32 engine diagnostics on the C5 twin (n = 2e4, m = 1024, d = 1024): the four Gram blocks of the fp32 engine against
the fp64 engine's (nk_nystrom_gram) for several flush intervals, then the fitted operators / predictions against the fp64
fit.  Separates arithmetic (errors that shrink with the flush interval) from wiring mistakes (errors that do not)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(flush):
    import nys_koop_lqr_amd as nk
    from nys_koop_lqr_amd import _lib
    g = np.load("tests/golden/f11_c5_twin.npz")
    n, d, p, m = int(g["n"]), int(g["d"]), int(g["p"]), int(g["m"])
    rng = np.random.default_rng(int(g["seed"]))
    S = rng.standard_normal((n, d)).astype(np.float32); U = rng.standard_normal((n, p)).astype(np.float32)
    Wt = (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d)).astype(np.float32); Bt = (rng.standard_normal((p, d)) * 0.1).astype(np.float32)
    Y = (np.tanh(S.astype(np.float64) @ Wt) + U.astype(np.float64) @ Bt).astype(np.float32).astype(np.float64)
    X = np.hstack([S, U]).astype(np.float64)
    ls, gamma = float(g["ls"]), float(g["gamma"])
    mp = m + p
    def mk(dtype):
        reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(ls, ls, ls, d), gamma=gamma, m=m)
        reg.compute_dtype = dtype
        reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, g["idx"]])
        return reg
    ctx = nk.get_context()
    out = {}
    grams = {}
    for dtype in ("f64", "f32"):
        reg = mk(dtype)
        ctx.set_compute_dtype(dtype)
        G = reg.gram_partial(X, Y)
        ctx.set_compute_dtype("f64")
        b1 = ((2 * m + p) * mp + 1) & ~1
        grams[dtype] = dict(G1=G[:mp * mp].reshape(mp, mp), G2=G[mp * mp:mp * mp + m * mp].reshape(m, mp),
                            G3=G[b1:b1 + m * m].reshape(m, m), G4=G[b1 + m * m:b1 + m * m + d * m].reshape(d, m))
    for k in ("G1", "G2", "G3", "G4"):
        a, b = grams["f32"][k], grams["f64"][k]
        e = a - b
        out[k] = dict(relF=float(np.linalg.norm(e) / np.linalg.norm(b)), max_abs=float(np.abs(e).max()), mean_err=float(e.mean()),
                      mean_val=float(b.mean()), spectral_err=float(np.linalg.norm(e, 2)) if k != "G4" else None)
    # K block / U block / K_out consistency of G1: the U x U corner is exact data
    out["G1_UU_relerr"] = float(np.linalg.norm(grams["f32"]["G1"][m:, m:] - grams["f64"]["G1"][m:, m:]) / np.linalg.norm(grams["f64"]["G1"][m:, m:]))
    regs = {}
    for dtype in ("f64", "f32"):
        reg = mk(dtype); reg.fit(X, Y); regs[dtype] = reg
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    out["ops_vs_f64"] = dict(A=rel(regs["f32"].A, regs["f64"].A), B=rel(regs["f32"].B, regs["f64"].B), C=rel(regs["f32"].C, regs["f64"].C),
                             predict=rel(regs["f32"].predict(X[g["q"]]), regs["f64"].predict(X[g["q"]])))
    out["ms"] = dict(f64=dict(kmat=regs["f64"].fit_stats_["ms_kmat"], gram=regs["f64"].fit_stats_["ms_gram"]),
                     f32=dict(kmat=regs["f32"].fit_stats_["ms_kmat"], gram=regs["f32"].fit_stats_["ms_gram"]))
    print("FLUSH", flush, json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:
        for fl in (32, 4, 1):
            env = dict(os.environ, NYSKOOP_F32_FLUSH=str(fl))
            r = subprocess.run([sys.executable, __file__, str(fl)], env=env, capture_output=True, text=True, timeout=600)
            print(r.stdout[-3000:], r.stderr[-1500:] if r.returncode else "")
