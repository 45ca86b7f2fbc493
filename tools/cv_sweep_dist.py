"""Sharded CV sweep on GPUs: run with torchrun; every rank fits its (candidate, fold) units with the HIP path
and one all-gather collects the scores.  NYSKOOP_BENCH_BACKEND=gloo lets several ranks share one GPU for rehearsal.
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/cv_sweep_dist.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nys_koop_lqr_amd as nk
from nys_koop_lqr_amd import dist as nkd
backend = os.environ.get("NYSKOOP_BENCH_BACKEND", "nccl")
if backend != "nccl":
    os.environ["NYSKOOP_DEVICE"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
rank, world = nkd.init_process_group(backend)
rng = np.random.default_rng(0)  # same data on every rank (replicated dataset)
n, d, p, m = 5050, 192, 6, 500
S = rng.standard_normal((n, d)); U = rng.standard_normal((n, p))
Y = np.tanh(S @ (rng.standard_normal((d, d)) * 0.9 / np.sqrt(d))) + U @ (rng.standard_normal((p, d)) * 0.1)
X = np.hstack([S, U])
cands = [dict(kernel=nk.ThreeDimensionalKernel(l, l, l, d), gamma=g, m=m) for l in (10., 20., 40.) for g in (1e-6, 1e-5, 1e-4, 1e-3)]
nkd.sharded_grid_search(X, Y, p, cands[:1], n_splits=5, seed=1)  # warm-up
import torch.distributed as td
if world > 1:
    td.barrier()
t0 = time.perf_counter()
res = nkd.sharded_grid_search(X, Y, p, cands, n_splits=5, seed=1)
if world > 1:
    td.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    print(f"world {world}: {len(cands) * 5} units in {dt:.3f} s = {len(cands) * 5 / dt:.1f} units/s; best candidate {res['best_index']} "
          f"mean scores {np.round(res['mean_test_score'], 5).tolist()}")
if world > 1:
    td.destroy_process_group()
