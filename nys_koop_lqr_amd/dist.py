"""Multi-GPU sharding of the hyper-parameter sweep: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm), the (candidate, fold) work list of learn_hyperparams (benchmark_lqr_cloth.py:39-66) is dealt
round-robin over the ranks, every rank fits its units on its own GPU against its own replica of the dataset, and
ONE all-gather collects the per-unit scores.  No collective touches the data path.
"""
import os

import numpy as np

from . import harness


def init_process_group(backend=None):
    """Rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return 0, 1  # plain `python script.py`: single process, no process group needed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("NYSKOOP_DEVICE", str(local))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_units(n_units, rank, world):
    """Round-robin deal of unit indices: rank r gets r, r+world, ... (equal cost per unit within a candidate)."""
    return list(range(rank, n_units, world))


def all_gather_scores(local_scores, n_units, rank, world):
    """One all-gather of fixed-size per-rank score vectors (NaN padded); returns the (n_units,) array in unit order."""
    import torch
    import torch.distributed as dist
    per = (n_units + world - 1) // world
    buf = np.full(per, np.nan)
    buf[: len(local_scores)] = local_scores
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    mine = torch.from_numpy(buf).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    out = np.full(n_units, np.nan)
    for r in range(world):
        idx = shard_units(n_units, r, world)
        out[idx] = gathered[r].cpu().numpy()[: len(idx)]
    return out


def sharded_grid_search(X, Y, n_inputs, candidates, n_splits=5, centers=None, unit_fn=None, seed=None, workers=1):
    """Distributed counterpart of harness.grid_search_cv.  Landmarks: `centers[(c, f)]` if given, otherwise drawn
    from a per-unit RandomState(seed + unit index) so that the result does not depend on the world size.
    Every rank returns the same dict (split_scores, mean_test_score, best_index, best_params)."""
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    unit_fn = unit_fn or harness.cv_unit_score
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    folds = harness.kfold_slices(X.shape[0], n_splits)
    units = harness.cv_work_list(len(candidates), n_splits)
    mine = shard_units(len(units), rank, world)
    def run(u):
        c, f = units[u]
        if centers is not None:
            idx = centers[(c, f)]
        else:
            n_train = X.shape[0] - (folds[f][1] - folds[f][0])
            rs = np.random.RandomState((0 if seed is None else int(seed)) + u)
            idx = rs.choice(np.arange(0, n_train), size=candidates[c]["m"], replace=False)
        return unit_fn(X, Y, n_inputs, candidates[c], folds[f], idx)

    if workers > 1 and len(mine) > 1:  # several latency-bound fits in flight per GPU (one context per thread)
        from ._lib import worker_pool
        local = list(worker_pool(workers).map(run, mine))
    else:
        local = [run(u) for u in mine]
    if world > 1:
        flat = all_gather_scores(np.asarray(local, dtype=np.float64), len(units), rank, world)
    else:
        flat = np.asarray(local, dtype=np.float64)
    scores = flat.reshape(len(candidates), n_splits)
    mean = scores.mean(axis=1)
    best = int(np.argmax(mean))
    return dict(split_scores=scores, mean_test_score=mean, best_index=best, best_params=candidates[best])
