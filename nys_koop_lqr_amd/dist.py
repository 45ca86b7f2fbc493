"""Multi-GPU sharding of the hyper-parameter sweep: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm), the (candidate, fold) work list of learn_hyperparams (benchmark_lqr_cloth.py:39-66) is dealt
round-robin over the ranks, every rank fits its units on its own GPU against its own replica of the dataset, and
ONE all-gather collects the per-unit scores.  No collective touches the data path of the sweep.
`sample_sharded_fit` is the other natural sharding: one large fit with the samples split over the ranks
and one all-reduce of the Gram accumulators as its only exchange step.
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


def sharded_grid_search(X, Y, n_inputs, candidates, n_splits=5, centers=None, unit_fn=None, seed=None, workers=1,
                        error_score=np.nan, batch=0, batch_groups=1):
    """Distributed counterpart of harness.grid_search_cv.  Landmarks: `centers[(c, f)]` if given, otherwise drawn
    from a per-unit RandomState(seed + unit index) so that the result does not depend on the world size.
    A unit whose fit fails numerically scores `error_score` (nan by default, like GridSearchCV) and the sweep goes on.
    batch / batch_groups: run each rank's share through the lock-step batched sweep (harness.grid_search_cv).
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
        if unit_fn is harness.cv_unit_score:
            return unit_fn(X, Y, n_inputs, candidates[c], folds[f], idx, error_score)
        return unit_fn(X, Y, n_inputs, candidates[c], folds[f], idx)

    if batch > 1 and unit_fn is harness.cv_unit_score and len(mine) > 1:
        # this rank's share through the lock-step batched sweep (nk_cv_grid): same scores, several thousand units/s
        def landmarks(u):
            c, f = units[u]
            if centers is not None:
                return centers[(c, f)]
            n_train = X.shape[0] - (folds[f][1] - folds[f][0])
            rs = np.random.RandomState((0 if seed is None else int(seed)) + u)
            return rs.choice(np.arange(0, n_train), size=candidates[c]["m"], replace=False)
        my_units = [units[u] for u in mine]
        res = harness.grid_search_cv(X, Y, n_inputs, candidates, n_splits=n_splits,
                                     centers={units[u]: landmarks(u) for u in mine}, work=my_units, batch=batch,
                                     batch_groups=batch_groups, error_score=error_score)
        local = [res["split_scores"][c, f] for (c, f) in my_units]
    elif workers > 1 and len(mine) > 1:  # several latency-bound fits in flight per GPU (one context per thread)
        from ._lib import worker_pool
        local = list(worker_pool(workers).map(run, mine))
    else:
        local = [run(u) for u in mine]
    if world > 1:
        flat = all_gather_scores(np.asarray(local, dtype=np.float64), len(units), rank, world)
    else:
        flat = np.asarray(local, dtype=np.float64)
    scores = flat.reshape(len(candidates), n_splits)
    mean, best = harness._rank_candidates(scores)  # a failed (NaN) candidate ranks last, as in GridSearchCV
    return dict(split_scores=scores, mean_test_score=mean, best_index=best,
                best_params=candidates[best] if best >= 0 else None)


def sample_sharded_fit(reg, X_local, Y_local, landmark_rows=None):
    """One LARGE fit over several GPUs: the samples are sharded, every rank holds all landmarks,
    accumulates the four Gram blocks of its own rows on its GPU (`reg.gram_partial`), ONE all-reduce sums the packed
    accumulators (102 MB at m=2000, d=384; RCCL on device memory with the nccl backend, no host copy), and every rank
    finishes the O(m^3) stage from the sum (`reg.fit_from_gram`), so all ranks end with the same fitted regressor.
    Summation order differs from the single-GPU fit: last-bit differences only.

    Landmarks: `reg.nystrom_centers_output` if already set (identical on all ranks); otherwise `landmark_rows`, GLOBAL
    row indices into the concatenation of the shards in rank order (default: rank 0 draws them like regressors.py:130
    from the global legacy RNG); the owning ranks contribute the rows and one small all-reduce assembles them.
    X_local: n_local x (d+p), Y_local: n_local x d (NumPy arrays or device tensors)."""
    import torch
    import torch.distributed as dist
    from . import _lib
    n_local = int(Y_local.shape[0])
    d = int(Y_local.shape[1])
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(), dist.get_world_size()
    # (a one-rank nccl group takes the device path too: RCCL initialises and reduces over one rank, so the branch the
    # multi-GPU runs use is exercised on a single-GPU box -- tests/test_gpu_round3.py)
    use_cuda = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = n_local
    if world > 1:
        dist.all_reduce(counts)
    counts = counts.cpu().numpy()
    n_total = int(counts.sum())
    first = int(counts[:rank].sum())
    if reg.nystrom_centers_output is None:
        m = int(reg.m)
        idx = torch.zeros(m, dtype=torch.int64, device=dev)
        if landmark_rows is not None:
            idx = torch.as_tensor(np.asarray(landmark_rows, dtype=np.int64), device=dev)
        elif rank == 0:
            idx = torch.from_numpy(np.random.choice(np.arange(0, n_total), size=m, replace=False).astype(np.int64)).to(dev)
        if world > 1 and landmark_rows is None:
            dist.broadcast(idx, src=0)
        idx = idx.cpu().numpy()
        Z = np.zeros((len(idx), d))
        mine = np.nonzero((idx >= first) & (idx < first + n_local))[0]
        if len(mine):
            loc = (idx[mine] - first).tolist()
            rows = Y_local[loc]
            Z[mine] = rows.cpu().numpy() if hasattr(rows, "cpu") else np.asarray(rows)
        zt = torch.from_numpy(Z).to(dev)
        if world > 1:
            dist.all_reduce(zt)  # every landmark row has exactly one owner, the others contribute zeros
        reg.nystrom_centers_output = np.ascontiguousarray(zt.cpu().numpy().T)
        reg.nystrom_centers_input = None
    cnt = reg.gram_size(d)
    if use_cuda:
        gram = torch.empty(cnt, dtype=torch.float64, device=dev)
    else:
        try:  # page-locked when a GPU is there (host-side collective, e.g. a gloo rehearsal on a GPU box)
            gram = torch.from_numpy(_lib.pinned_empty((cnt,)))
        except Exception:
            gram = torch.empty(cnt, dtype=torch.float64)
    # Stream ordering (device path): gram_partial returns with the accumulator complete (the library synchronises its
    # stream before returning); the all-reduce runs on RCCL's stream, which torch orders after its current stream, and
    # `fit_from_gram` orders the library's (non-blocking) streams after torch's current stream (nk_wait_stream through
    # Context.wait_for) -- and torch's current stream after the collective through work.wait() -- before it reads the sum.
    reg.gram_partial(X_local, Y_local, out=gram if use_cuda else gram.numpy())
    if world > 1 or use_cuda:
        work = dist.all_reduce(gram, async_op=True)
        work.wait()  # nccl: torch's current stream now waits for the collective; gloo: blocks until done
    reg.fit_from_gram(gram if use_cuda else gram.numpy(), n_total, d)
    return reg
