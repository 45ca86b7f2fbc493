#!/usr/bin/env python3
"""Where does the operator error of an ill-conditioned fit come from?  Config 2 (f8: HJB N = 1e4, m = 200, Matern-5/2,
cond(inner) = 1.3e13): the GPU's Gram blocks against NumPy's, then the fit emulated in NumPy / SciPy FROM THE GPU'S GRAM BLOCKS
(Cholesky solves, eigen square root: the GPU's algebra with LAPACK's arithmetic) against the GPU's operators and the golden."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg
import nys_koop_lqr_amd as nk
from oracle import nk_oracle as O
g = np.load("tests/golden/f8_hjb_config2.npz")
X, Y, idx = g["X"], g["Y"], g["idx"]
ls, gamma, m = float(g["ls"]), float(g["gamma"]), int(g["m"])
n, d, p = X.shape[0], 1, 1
mp = m + p
relf = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
reg = nk.KoopmanNystromRegressor(p, kernel=nk.KernelWrapper([ls]), gamma=gamma, m=m)
reg.nystrom_centers_output = np.ascontiguousarray(Y.T[:, idx])
G = reg.gram_partial(X, Y)
b1 = ((2 * m + p) * mp + 1) & ~1
G1, G2 = G[:mp * mp].reshape(mp, mp), G[mp * mp:mp * mp + m * mp].reshape(m, mp)
G3, G4 = G[b1:b1 + m * m].reshape(m, m), G[b1 + m * m:b1 + m * m + d * m].reshape(d, m)
ok = O.KernelWrapper([ls]).kernel
Z = Y[idx]
Pin = np.hstack([ok(X[:, :d], Z), X[:, d:]]); Pout = ok(Y, Z)
R1, R2, R3, R4 = Pin.T @ Pin, Pout.T @ Pin, Pout.T @ Pout, Y.T @ Pout
print("Gram blocks, GPU vs NumPy: relF", [relf(a, b) for a, b in ((G1, R1), (G2, R2), (G3, R3), (G4, R4))],
      "max rel entry", [float(np.max(np.abs(a - b) / np.abs(b))) for a, b in ((G1, R1), (G2, R2), (G3, R3), (G4, R4))])
Kg = np.empty((m, m)); Kc = ok(Z, Z)
Kg = nk.KernelWrapper([ls]).kernel(Z, Z)
print("K_mm GPU vs NumPy: max abs", float(np.abs(Kg - Kc).max()), "features K(X,Z): max abs",
      float(np.abs(nk.KernelWrapper([ls]).kernel(X[:, :d], Z) - Pin[:, :m]).max()))

def emulate(G1, G2, G3, G4, K):
    Kj = K + 1e-6 * np.eye(m)
    w, V = np.linalg.eigh(Kj); S = (V * np.sqrt(w)) @ V.T; Sinv = (V / np.sqrt(w)) @ V.T
    inner = G1 + gamma * n * scipy.linalg.block_diag(Kj, np.eye(p)); inner_rec = gamma * n * Kj + G3
    Vs = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner), G2.T).T
    Gm = Sinv @ (Vs @ scipy.linalg.block_diag(K @ Sinv, np.eye(p)))
    C = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner_rec), G4.T).T @ S
    return Gm[:, :m], Gm[:, m:], C
reg.fit(X, Y)
for name, blocks in (("NumPy Gram blocks", (R1, R2, R3, R4, Kc)), ("GPU Gram blocks", (G1, G2, G3, G4, Kg))):
    A, B, C = emulate(*blocks)
    print(f"emulated fit from {name}: vs golden A {relf(A, g['A']):.2e} B {relf(B, g['B']):.2e} C {relf(C, g['C']):.2e} | "
          f"GPU fit vs this emulation A {relf(reg.A, A):.2e} B {relf(reg.B, B):.2e} C {relf(reg.C, C):.2e}")
print(f"GPU fit vs golden: A {relf(reg.A, g['A']):.2e} B {relf(reg.B, g['B']):.2e} C {relf(reg.C, g['C']):.2e}")

# ---- the O(m^3) building blocks alone: backward errors of the GPU's SPD solve / square root against LAPACK's ----------------
import ctypes as C
from nys_koop_lqr_amd.regressors import KoopmanKernelRegressor as KK
from nys_koop_lqr_amd import _lib
ctx = nk.get_context()
Kj = Kc + 1e-6 * np.eye(m)
inner = R1 + gamma * n * scipy.linalg.block_diag(Kj, np.eye(p))
rhs = R2.T.copy()                       # (m+p) x m : cross^T
Xg = KK._solve_spd(ctx, inner, rhs)
Xl = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner), rhs)
be = lambda Xs: float(np.linalg.norm(inner @ Xs - rhs) / (np.linalg.norm(inner) * np.linalg.norm(Xs)))
print(f"solve inner X = cross^T: backward error GPU {be(Xg):.2e} LAPACK {be(Xl):.2e}; GPU vs LAPACK solution relF {relf(Xg, Xl):.2e}")
S = np.empty((m, m)); Si = np.empty((m, m)); it = C.c_int32(); res = C.c_double()
_lib.check(ctx.lib.nk_sqrtm_spd(ctx.handle, Kj.ctypes.data, m, m, S.ctypes.data, Si.ctypes.data, C.byref(it), C.byref(res)))
w, V = np.linalg.eigh(Kj); Sl = (V * np.sqrt(w)) @ V.T; Sil = (V / np.sqrt(w)) @ V.T
print(f"sqrt of K_mm + jitter (cond {w[-1] / w[0]:.1e}): GPU ||S S - K||/||K|| {relf(S @ S, Kj):.2e} (LAPACK eigh {relf(Sl @ Sl, Kj):.2e}); "
      f"S vs eigh {relf(S, Sl):.2e}; S^-1 vs eigh {relf(Si, Sil):.2e}; ||S^-1 S - I|| GPU {np.linalg.norm(Si @ S - np.eye(m)):.2e} eigh {np.linalg.norm(Sil @ Sl - np.eye(m)):.2e}")
# the A formula with mixed ingredients: which GPU ingredient costs the accuracy?
Vs_l = Xl.T; Vs_g = Xg.T
mk = lambda Vs, Sinv_: (Sinv_ @ (Vs @ scipy.linalg.block_diag(Kc @ Sinv_, np.eye(p))))[:, :m]
Aref = mk(Vs_l, Sil)
print(f"A from (GPU solve, LAPACK sqrt) vs all-LAPACK: {relf(mk(Vs_g, Sil), Aref):.2e}; from (LAPACK solve, GPU sqrt): {relf(mk(Vs_l, Si), Aref):.2e}; "
      f"(GPU, GPU): {relf(mk(Vs_g, Si), Aref):.2e}; GPU fit's A vs all-LAPACK: {relf(reg.A, Aref):.2e}")
