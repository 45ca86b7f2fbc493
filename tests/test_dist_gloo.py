"""CPU, world_size 2, gloo: the sharded CV sweep deals (candidate, fold) units round-robin, all-gathers the scores and
every rank ends with the same table as the serial sweep.  The per-unit fit is injected (the oracle) because no GPU is
present here; on the GPU box the default unit function runs the HIP fit."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, {root!r})
    from nys_koop_lqr_amd import dist as nkd, harness
    from oracle import nk_oracle as O
    rank, world = nkd.init_process_group("gloo")
    g = dict(np.load(os.path.join({root!r}, "tests", "golden", "f5_cloth_gridsearch.npz")))
    X, Y = g["X"][:, :], g["Y"]
    cands = [dict(kernel=tuple(g["cands"][k]), gamma=float(gm), m=12) for k in (0, 1) for gm in (1e-5, 1e-3)]
    def unit(X, Y, p, params, fold, idx):
        mk = lambda: O.KoopmanNystromOracle(p, kernel=O.ThreeDimensionalKernel(*params["kernel"], 192),
                                            gamma=params["gamma"], m=params["m"], faithful=False)
        return O.cv_fold_score(mk, X, Y, fold, idx)
    res = nkd.sharded_grid_search(X, Y, 6, cands, n_splits=5, unit_fn=unit, seed=7)
    assert nkd.shard_units(20, rank, world) == list(range(rank, 20, world))
    np.save(os.path.join({out!r}, f"scores_{{rank}}.npy"), res["split_scores"])
    print(json.dumps(dict(rank=rank, world=world, best=res["best_index"])))
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sharded_grid_search_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    s0 = np.load(tmp_path / "scores_0.npy")
    s1 = np.load(tmp_path / "scores_1.npy")
    assert s0.shape == (4, 5) and np.array_equal(s0, s1) and np.all(np.isfinite(s0))
    # serial sweep (world size 1, same per-unit seeds) gives the same table
    sys.path.insert(0, ROOT)
    from nys_koop_lqr_amd import dist as nkd
    from oracle import nk_oracle as O
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "f5_cloth_gridsearch.npz")))
    cands = [dict(kernel=tuple(g["cands"][k]), gamma=float(gm), m=12) for k in (0, 1) for gm in (1e-5, 1e-3)]

    def unit(X, Y, p, params, fold, idx):
        mk = lambda: O.KoopmanNystromOracle(p, kernel=O.ThreeDimensionalKernel(*params["kernel"], 192),
                                            gamma=params["gamma"], m=params["m"], faithful=False)
        return O.cv_fold_score(mk, X, Y, fold, idx)
    serial = nkd.sharded_grid_search(g["X"], g["Y"], 6, cands, n_splits=5, unit_fn=unit, seed=7)
    assert np.allclose(serial["split_scores"], s0, rtol=1e-12, atol=0)
