"""GPU tests added in round 3: error propagation out of a merged lock-step flush, the slow-path counters, and the
software-pipelined TN engine (results against NumPy on shapes that hit every tile class)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import relf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nk():
    import nys_koop_lqr_amd as nk
    nk.get_context()
    return nk


def test_failed_merged_launch_reaches_every_unit_of_the_round():
    """A merged launch is issued by whichever member thread arrives last; HIP's last error is per host thread.  The first
    error of a flush must reach EVERY member the flush releases: with the 40th merged launch of the process made invalid
    (block of 4096 threads), all units of that round report a non-zero status and a NaN score -- none reports success
    on buffers the device never wrote -- and a later sweep in the same process is clean again."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import _lib
rng = np.random.default_rng(3)
n, d, p, m = 505, 24, 2, 64
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.8 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
kern = nk.ThreeDimensionalKernel(4., 4., 4., d).kernel
folds = [(0, 101), (101, 202), (202, 303), (303, 404), (404, 505)]
units = []
for u in range(8):
    lo, hi = folds[u %% 5]
    idx = np.random.RandomState(u).choice(n - 101, m, replace=False)
    units.append((kern, 1e-4, 1e-6, m, (lo, hi), np.where(idx < lo, idx, idx + (hi - lo))))
pool = _lib.lockstep_pool(8)
sc, st = pool.cv_grid(X, Y, p, units)
print("ROUND1", int(np.count_nonzero(st)), int(np.count_nonzero(np.isfinite(sc))))
sc2, st2 = pool.cv_grid(X, Y, p, units)
print("ROUND2", int(np.count_nonzero(st2)), int(np.count_nonzero(np.isfinite(sc2))))
''' % ROOT
    env = dict(os.environ, NYSKOOP_GROUP_TEST_FAIL_MERGED="40")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
    lines = dict(l.split(" ", 1) for l in out.stdout.splitlines() if l.startswith("ROUND"))
    bad1, finite1 = (int(v) for v in lines["ROUND1"].split())
    assert bad1 == 8 and finite1 == 0, lines    # every unit of the failing round saw the error
    bad2, finite2 = (int(v) for v in lines["ROUND2"].split())
    assert bad2 == 0 and finite2 == 8, lines    # and the group is usable afterwards


def test_runtime_counters_count_the_rank_truncating_branch(nk):
    from nys_koop_lqr_amd import _lib
    before = _lib.runtime_counters()
    rng = np.random.default_rng(5)
    n, d, p, m = 300, 6, 1, 24
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S) + 0.1 * U
    X = np.hstack([S, U])
    reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(2., 2., 2., d), gamma=1e-14, m=m)
    idx = np.concatenate([np.arange(16), np.arange(8)])  # duplicated landmarks: an exact null space
    reg.nystrom_centers_output = Y.T[:, idx]
    reg.fit(X, Y)
    after = _lib.runtime_counters()
    assert after["rank_truncated_fits"] == before["rank_truncated_fits"] + 1
    assert set(after) == {"chain_giveups", "jacobi_giveups", "rank_truncated_fits", "sqrt_retries", "refined_fits"}


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (2006, 2000, 1000), (384, 2000, 777), (130, 258, 17), (2, 2, 3),
                                   (640, 640, 5000)])
def test_tn_engine_products_vs_numpy(nk, M, N, K):
    """C = A^T B through nk_gemm (the LDS-DMA / MFMA engine with the software-pipelined operand fetch): full tiles, edge
    tiles in both directions, K tails that are not multiples of the 16-row step, a K range shorter than one step."""
    from nys_koop_lqr_amd import _lib
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    A = rng.standard_normal((K, M)); B = rng.standard_normal((K, N))
    ctx = nk.get_context()
    C = np.empty((M, N))
    _lib.check(ctx.lib.nk_gemm(ctx.handle, 1, 0, M, N, K, 1.0, A.ctypes.data, M, B.ctypes.data, N, 0.0, C.ctypes.data, N))
    ref = A.T @ B
    assert relf(C, ref) < 5e-15


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (2006, 2000, 4100), (384, 2000, 777), (132, 260, 33), (4, 4, 5)])
def test_fp32_engine_gram_products_vs_numpy(nk, M, N, K):
    """C (fp64) = A^T B with fp32 operands through the fp32 engine's Gram entry (nk_bench-free path: a fit-shaped call is
    not needed, the engine is reachable through nk_gemm_f32): products of fp32 numbers accumulated in fp32 for at most
    1024 rows at a time, then in fp64 -- compared with the fp64 product of the same rounded operands."""
    from nys_koop_lqr_amd import _lib
    rng = np.random.default_rng(M + 3 * N + 11 * K)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    lda, ldb = (M + 3) & ~3, (N + 3) & ~3
    Ap = np.zeros((K, lda), dtype=np.float32); Ap[:, :M] = A
    Bp = np.zeros((K, ldb), dtype=np.float32); Bp[:, :N] = B
    ctx = nk.get_context()
    C = np.empty((M, N))
    _lib.check(ctx.lib.nk_gemm_f32(ctx.handle, M, N, K, Ap.ctypes.data, lda, Bp.ctypes.data, ldb, C.ctypes.data, N))
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    err = np.abs(C - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (128, 128, 64), (132, 260, 96), (200, 72, 160), (384, 2000, 800),
                                   (2006, 2000, 4096), (2006, 2000, 12288)])
def test_fp32_engine_assembly_k_loop_gives_the_bits_of_the_compiled_loop(nk, monkeypatch, M, N, K):
    """The fp32 Gram launches with the whole k loop as one hand-scheduled assembly block (accumulators double buffered in the
    accumulation registers, the fp64 flush in the gaps between the matrix instructions; K a multiple of 32) against the
    compiler-scheduled kernel (NYSKOOP_F32_ASM=0, read per launch): same products, same flush order -- the same bits; K from
    one step (final step only) over odd and even step counts to split-K ranges."""
    from nys_koop_lqr_amd import _lib
    rng = np.random.default_rng(M + 3 * N + 11 * K)
    lda, ldb = (M + 3) & ~3, (N + 3) & ~3
    Ap = np.zeros((K, lda), dtype=np.float32); Ap[:, :M] = rng.standard_normal((K, M)).astype(np.float32)
    Bp = np.zeros((K, ldb), dtype=np.float32); Bp[:, :N] = rng.standard_normal((K, N)).astype(np.float32)
    ctx = nk.get_context()

    def product():
        C = np.empty((M, N))
        _lib.check(ctx.lib.nk_gemm_f32(ctx.handle, M, N, K, Ap.ctypes.data, lda, Bp.ctypes.data, ldb, C.ctypes.data, N))
        return C
    Ca = product()
    monkeypatch.setenv("NYSKOOP_F32_ASM", "0")
    Cc = product()
    monkeypatch.delenv("NYSKOOP_F32_ASM")
    ref = Ap[:, :M].astype(np.float64).T @ Bp[:, :N].astype(np.float64)
    assert np.abs(Cc - ref).max() / np.abs(ref).max() < 2e-6
    if K <= 4096:
        assert np.array_equal(Ca, Cc), float(np.abs(Ca - Cc).max())
    else:  # the two kernels split K differently (one against two workgroups per CU): the fp64 sums of the slices differ in order
        assert np.abs(Ca - Cc).max() / np.abs(ref).max() < 1e-15


def test_fp32_engine_fit_with_assembly_k_loops_gives_the_bits_of_the_compiled_loops(nk, monkeypatch):
    """A whole fit on the fp32 engine (kernel blocks and Gram launches through the assembly k loops: d and the row count multiples
    of 32) against the same fit with NYSKOOP_F32_ASM=0: identical operators."""
    rng = np.random.default_rng(31)
    n, d, p, m = 4096, 64, 3, 256
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    X = np.hstack([S, U])

    def fit():
        reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(6., 6., 6., d), gamma=1e-4, m=m)
        reg.compute_dtype = "f32"
        reg.nystrom_centers_output = np.ascontiguousarray(Y[:m].T)
        reg.fit(X, Y)
        return [np.array(a) for a in (reg.A, reg.B, reg.C)]
    asm = fit()
    monkeypatch.setenv("NYSKOOP_F32_ASM", "0")
    compiled = fit()
    monkeypatch.delenv("NYSKOOP_F32_ASM")
    for a, b in zip(asm, compiled):
        assert np.all(np.isfinite(a)) and np.array_equal(a, b)


def test_cholesky_lookahead_gives_the_same_bits(nk, monkeypatch):
    """The blocked Cholesky with look-ahead (next block column on the chain's stream, the rest of the trailing update on a
    second stream) performs exactly the arithmetic of the sequential order: a fit with it computes the same bits as a fit
    without (NYSKOOP_CHOL_LOOKAHEAD=0 is read once per process, so the two fits run in child processes)."""
    code = r'''
import numpy as np, sys, hashlib
sys.path.insert(0, %r)
import nys_koop_lqr_amd as nk
rng = np.random.default_rng(8)
n, d, p, m = 3000, 40, 3, 700
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(6., 6., 6., d), gamma=1e-5, m=m)
reg.nystrom_centers_output = np.ascontiguousarray(Y[:m].T)
reg.fit(X, Y)
h = hashlib.sha256(np.ascontiguousarray(reg.A).tobytes() + np.ascontiguousarray(reg.B).tobytes() + np.ascontiguousarray(reg.C).tobytes()).hexdigest()
print("HASH", h)
''' % ROOT
    hashes = []
    for la in ("1", "0"):  # (off by default: nk_linalg.hip)
        env = dict(os.environ, NYSKOOP_CHOL_LOOKAHEAD=la)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, (out.stdout[-300:], out.stderr[-1500:])
        hashes.append([l for l in out.stdout.splitlines() if l.startswith("HASH")][0])
    assert hashes[0] == hashes[1]


def test_sample_sharded_fit_through_rccl_with_one_rank():
    """The device branch of dist.sample_sharded_fit (packed Gram accumulator in HBM -> RCCL all-reduce on torch's stream
    -> nk_wait_stream -> nk_nystrom_solve) with backend 'nccl' and ONE rank: RCCL initialises, the collective runs on
    the device buffer, and the operators equal those of the plain fit bit for bit (one shard: same summation order).
    The multi-rank runs are the driver's; this keeps the branch they take executed on the one-GPU box."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"; os.environ["LOCAL_RANK"] = "0"
import torch, torch.distributed as dist
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import dist as nkd
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
rng = np.random.default_rng(2)
n, d, p, m = 4000, 48, 3, 256
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
mk = lambda: nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(6., 6., 6., d), gamma=1e-5, m=m)
a = mk(); a.nystrom_centers_output = np.ascontiguousarray(Y[:m].T); a.fit(X, Y)
b = mk(); b.nystrom_centers_output = np.ascontiguousarray(Y[:m].T)
Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
nkd.sample_sharded_fit(b, Xd, Yd)
print("EQUAL", bool(np.array_equal(a.A, b.A) and np.array_equal(a.B, b.B) and np.array_equal(a.C, b.C)), dist.get_backend())
dist.destroy_process_group()
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-300:], out.stderr[-2000:])
    assert "EQUAL True nccl" in out.stdout, out.stdout[-300:]


def test_optional_refinement_with_doubled_precision_residuals(nk, golden):
    """nk_set_refine: an ill-conditioned regularised system (config 2, cond 1.3e13) refined with residuals accumulated in
    doubled precision -- the corrections contract (first one 4e-4 of the solution: the blocked solve is already backward
    stable), both steps are applied to both systems, the operators stay within the reference's own reproducibility (f8b) and
    move towards the reference; a well-conditioned fit is left alone; switched off again the bits are those of before."""
    from nys_koop_lqr_amd import _lib
    ctx = _lib.get_context()
    g = golden("f8_hjb_config2.npz")
    env = golden("f8b_hjb_envelope.npz")
    X, Y, idx = g["X"], g["Y"], g["idx"]

    def fit():
        reg = nk.KoopmanNystromRegressor(1, kernel=nk.KernelWrapper([float(g["ls"])]), gamma=float(g["gamma"]), m=200)
        reg.nystrom_centers_output = Y.T[:, idx]
        reg.fit(X, Y)
        return reg
    base = fit()
    assert base.fit_stats_["refined"] == 0
    ctx.set_refine(1e-9, 2)
    try:
        ref = fit()
        st = ref.fit_stats_
        assert st["refined"] == 2 + 16 * 2, st
        assert 0.0 < st["refine_ratio_inner"] < 1e-2 and 0.0 < st["refine_ratio_inner_rec"] < 1e-2, st
        bars = 3.0 * np.maximum(env["op_roworder"], env["op_envelope"])
        for nm, got, bar in (("A", ref.A, bars[0]), ("B", ref.B, bars[1]), ("C", ref.C, bars[2])):
            assert relf(got, g[nm]) < bar, (nm, relf(got, g[nm]), bar)
        assert relf(ref.A, g["A"]) <= relf(base.A, g["A"]) * 1.05
        print("\n[refinement] A vs reference:", relf(base.A, g["A"]), "->", relf(ref.A, g["A"]), "first corrections:",
              st["refine_ratio_inner"], st["refine_ratio_inner_rec"])
        # pivot ratio 9.5e-5 (f2): not touched
        g2 = golden("f2_synth_rbf_d384.npz")
        X2, Y2 = g2["X"].astype(np.float64), g2["Y"].astype(np.float64)
        l3 = g2["ls"] if g2["ls"].size == 3 else np.repeat(g2["ls"], 3)
        r2 = nk.KoopmanNystromRegressor(6, kernel=nk.ThreeDimensionalKernel(*l3, Y2.shape[1]), gamma=float(g2["gamma"]), m=len(g2["idx"]))
        r2.nystrom_centers_output = Y2.T[:, g2["idx"]]
        r2.fit(X2, Y2)
        assert r2.fit_stats_["refined"] == 0
    finally:
        ctx.set_refine(0.0, 2)
    again = fit()
    assert again.fit_stats_["refined"] == 0 and np.array_equal(again.A, base.A) and np.array_equal(again.C, base.C)


def test_blocked_solve_is_backward_stable_on_an_ill_conditioned_system(nk, golden):
    """The O(m^3) building block alone (nk_solve_spd: blocked Cholesky, products with inverted 64 x 64 diagonal blocks + one
    correction step each) on config 2's regularised system (cond 1.3e13, first diagonal block of the factor: cond 2e6): the
    normwise backward error is at LAPACK's level (round 2: 1.4e-15 = 20 x LAPACK's), and the forward distance to LAPACK's
    solution is what two backward-stable solvers differ by."""
    import scipy.linalg
    from oracle import nk_oracle as O
    from nys_koop_lqr_amd import _lib
    from nys_koop_lqr_amd.regressors import KoopmanKernelRegressor as KK
    g = golden("f8_hjb_config2.npz")
    X, Y, idx = g["X"], g["Y"], g["idx"]
    n, m, p, gamma = X.shape[0], int(g["m"]), 1, float(g["gamma"])
    k = O.KernelWrapper([float(g["ls"])]).kernel
    Z = Y[idx]
    Pin = np.hstack([k(X[:, :1], Z), X[:, 1:]])
    inner = Pin.T @ Pin + gamma * n * scipy.linalg.block_diag(k(Z, Z) + 1e-6 * np.eye(m), np.eye(p))
    rhs = (k(Y, Z).T @ Pin).T.copy()
    Xg = KK._solve_spd(_lib.get_context(), inner, rhs)
    Xl = scipy.linalg.cho_solve(scipy.linalg.cho_factor(inner), rhs)
    be = lambda Xs: float(np.linalg.norm(inner @ Xs - rhs) / (np.linalg.norm(inner) * np.linalg.norm(Xs)))
    print(f"\n[blocked solve, cond {np.linalg.cond(inner):.1e}] backward error GPU {be(Xg):.2e} LAPACK {be(Xl):.2e}; "
          f"GPU vs LAPACK {relf(Xg, Xl):.2e}")
    assert be(Xg) < 2.0 * be(Xl) and be(Xg) < 2e-16
    assert relf(Xg, Xl) < 5e-3   # cond x eps = 3e-3: the forward error either of them may carry


def test_fused_trailing_update_and_next_diagonal_block_give_the_same_bits(nk, golden, monkeypatch):
    """chol_trail_potrf_kernel (the trailing update of a block step and the factorisation of the next diagonal block in one
    launch, by the wave that owns that block) performs the arithmetic of the two separate launches: fits and the stand-alone
    solve compute the same bits with NYSKOOP_CHOL_FUSE=0 (read per call).  Shapes: several blocks with a short last one and
    an odd number of right-hand-side rows (m = 700, p = 3), a single block (f4: m = 40), and the CV sweep's lock-step form."""
    from nys_koop_lqr_amd import harness
    rng = np.random.default_rng(8)
    n, d, p, m = 3000, 40, 3, 700
    S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
    Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
    X = np.hstack([S, U])
    g4 = golden("f4_hjb_matern.npz")

    def run():
        reg = nk.KoopmanNystromRegressor(p, kernel=nk.ThreeDimensionalKernel(6., 6., 6., d), gamma=1e-5, m=m)
        reg.nystrom_centers_output = np.ascontiguousarray(Y[:m].T)
        reg.fit(X, Y)
        r4 = nk.KoopmanNystromRegressor(1, kernel=nk.KernelWrapper(g4["ls"]), gamma=float(g4["gamma"]), m=len(g4["idx"]))
        r4.nystrom_centers_output = g4["Y"].astype(np.float64).T[:, g4["idx"]]
        r4.fit(g4["X"].astype(np.float64), g4["Y"].astype(np.float64))
        cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, d), gamma=1e-5, m=200) for l in (4.0, 8.0)]
        np.random.seed(3)
        cv = harness.grid_search_cv(X[:1000], Y[:1000], p, cands, n_splits=5, batch=5, batch_groups=2)
        return [np.array(a) for a in (reg.A, reg.B, reg.C, r4.A, r4.C, cv["split_scores"])]
    fused = run()
    monkeypatch.setenv("NYSKOOP_CHOL_FUSE", "0")
    apart = run()
    monkeypatch.delenv("NYSKOOP_CHOL_FUSE")
    for a, b in zip(fused, apart):
        assert np.array_equal(a, b)
