"""CPU: the C-ABI library loads and exports every symbol include/nyskoop.h declares, the host classes mirror the
reference's estimator protocol, and nothing silently falls back to the CPU when no GPU is present."""
import os
import pickle
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from nys_koop_lqr_amd import _lib
    return _lib.load_library()


def test_library_exports_every_declared_symbol(lib):
    from nys_koop_lqr_amd import _lib
    header = open(os.path.join(ROOT, "include", "nyskoop.h")).read()
    declared = set(re.findall(r"\b(nk_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes found in the header"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nk_version() == 2


def test_no_cpu_fallback(lib):
    import nys_koop_lqr_amd as nk
    from nys_koop_lqr_amd import _lib
    if lib.nk_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.NyskoopError, match="no CPU fallback"):
        nk.get_context(0)
    reg = nk.KoopmanNystromRegressor(1, kernel=nk.KernelWrapper([1.0, 1.0]), gamma=1e-6, m=4)
    with pytest.raises(_lib.NyskoopError):
        reg.fit(np.zeros((10, 3)), np.zeros((10, 2)))
    with pytest.raises(_lib.NyskoopError):
        nk.KernelWrapper([1.0]).kernel(np.zeros((2, 1)), np.zeros((2, 1)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nys_koop_lqr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("nk_oracle-free", ""), os.path.join(dirpath, f)


def test_estimator_protocol_matches_reference():
    from sklearn.base import clone
    import nys_koop_lqr_amd as nk
    k = nk.ThreeDimensionalKernel(1, 10, 100, 192)
    assert k.kernel.length_scale.shape == (192,) and list(k.kernel.length_scale[:4]) == [1, 10, 100, 1]
    reg = nk.KoopmanNystromRegressor(6, kernel=k, gamma=1e-5, m=500)
    assert set(reg.get_params()) == {"n_inputs", "kernel", "gamma", "m"}  # regressors.py:115
    for attr in ("A", "B", "C", "weights", "nystrom_centers_input", "nystrom_centers_output"):
        assert getattr(reg, attr) is None
    assert reg.jitter == 1e-6
    reg.nystrom_centers_output = np.zeros((192, 500))
    c = clone(reg)
    assert c.nystrom_centers_output is None and c.m == 500 and c.kernel is not reg.kernel
    reg.set_params(gamma=1e-7)
    assert reg.gamma == 1e-7
    r2 = pickle.loads(pickle.dumps(reg))
    assert r2.gamma == 1e-7 and r2._model is None and r2.nystrom_centers_output.shape == (192, 500)
    base = nk.KoopmanRegressor(2, 1e-3)
    with pytest.raises(NotImplementedError):
        base.fit(None, None)
    with pytest.raises(NotImplementedError):
        base.lift(None)


def test_harness_bookkeeping():
    from nys_koop_lqr_amd import harness
    from oracle import nk_oracle as O
    for n in (10, 404, 1010, 4000):
        assert harness.kfold_slices(n, 5) == O.kfold_slices(n, 5)
    from sklearn.model_selection import KFold, ParameterGrid
    for n in (11, 404):
        sk = [(int(te[0]), int(te[-1]) + 1) for _, te in KFold(5).split(np.zeros(n))]
        assert harness.kfold_slices(n, 5) == sk
    grid = {"kernel": ["k0", "k1", "k2"], "gamma": [1e-7, 1e-6], "m": [500]}
    assert harness.parameter_grid(grid) == list(ParameterGrid(grid))
    assert harness.cv_work_list(2, 3) == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
    trajs = [np.arange(12.0).reshape(3, 4), np.arange(12.0).reshape(3, 4) + 100]
    ctrls = [np.ones((2, 4)), 2 * np.ones((2, 4))]
    X, Y = harness.create_data_matrices(trajs, ctrls, [0, 1])
    assert X.shape == (5, 6) and Y.shape == (3, 6) and Y[0, 0] == 1.0 and X[0, 3] == 100.0 and X[3, 5] == 2.0


def test_host_dlqr_matches_shipped_gain(golden):
    """The host DARE against the reference's shipped K_lqr (through the oracle's faithful operators)."""
    from nys_koop_lqr_amd.lqr import cloth_gain_for_simulator, dlqr
    from oracle import nk_oracle as O
    from conftest import relf
    g = golden("f6_cloth_known_gain.npz")
    tr, u = g["trajs"], g["inputs"]
    X = np.hstack([np.vstack((tr[i][:, :-1], u[i][:, :-1])) for i in range(30)]).T
    Y = np.hstack([tr[i][:, 1:] for i in range(30)]).T
    np.random.seed(1)
    reg = O.KoopmanNystromOracle(6, kernel=O.ThreeDimensionalKernel(10, 10, 10, 192), gamma=1e-7, m=100)
    reg.fit(np.ascontiguousarray(X), np.ascontiguousarray(Y))
    Q = 0.005 * reg.C.T @ reg.C
    K, _ = dlqr(reg.A, reg.B, (Q + Q.T) / 2, np.eye(6))
    assert relf(cloth_gain_for_simulator(K), g["K_lqr_seed_1"]) < 5e-3


def test_every_environment_switch_is_documented():
    """Every NYSKOOP_* variable the library or the package reads is listed in INTEGRATION.md (section 6)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for path in glob.glob(os.path.join(root, "nys_koop_lqr_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "nys_koop_lqr_amd", "*.py")):
        names.update(re.findall(r"NYSKOOP_[A-Z0-9_]+", open(path).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc and n not in ("NYSKOOP_BENCH_BACKEND",))
    assert not missing, f"undocumented environment switches: {missing}"


def test_no_raw_hip_stream_call_escapes_the_lockstep_wrappers():
    """nk_lockstep.h redefines hipLaunchKernelGGL / hipMemcpyAsync / hipStreamSynchronize / ... by macro so that a member of
    a lock-step group records instead of issuing.  Spellings the macros do not catch (<<< >>>, hipLaunchKernel, the
    blocking hipMemcpy / hipMemset family, hipDeviceSynchronize, graph or cooperative launches) would silently escape
    the recording: the only ones allowed are listed here with the reason they are safe."""
    allowed = {
        # nk_cv_grid stages X / Y once, before any member thread exists (blocking copies on the caller's thread)
        ("nk_api.hip", "hipMemcpy2D("): 3,
        # one-time zero page of a context, blocking on purpose (both streams read it)
        ("nk_gemm_tn.hip", "hipMemset("): 1,
    }
    pat = re.compile(r"<<<|\bhipLaunchKernel\(|\bhipMemcpy\(|\bhipMemcpy2D\(|\bhipMemset\(|\bhipDeviceSynchronize\(|"
                     r"\bhipModuleLaunchKernel\(|\bhipLaunchCooperativeKernel\(|\bhipGraphLaunch\(|\bhipMemcpyDtoH\(|"
                     r"\bhipMemcpyHtoD\(|\bhipMemsetD8\(")
    csrc = os.path.join(ROOT, "nys_koop_lqr_amd", "csrc")
    seen = {}
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")) or f in ("nk_group.hip", "nk_lockstep.h"):
            continue  # the two files that implement the wrappers call the real functions
        for line in open(os.path.join(csrc, f)):
            code = line.split("//")[0]
            for mt in pat.finditer(code):
                seen[(f, mt.group(0))] = seen.get((f, mt.group(0)), 0) + 1
    assert seen == allowed, f"raw HIP calls outside the lock-step wrappers: {seen} (allowed: {allowed})"
    # and every translation unit with device code sees the macros
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip") and f != "nk_group.hip":
            assert '#include "nk_common.h"' in open(os.path.join(csrc, f)).read(), f
    assert '#include "nk_lockstep.h"' in open(os.path.join(csrc, "nk_common.h")).read()
