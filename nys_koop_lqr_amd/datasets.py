"""Readers for the reference's on-disk data formats, so that configs 1 and 3 run end to end on the real
data: the cloth CSV triplets (benchmark_lqr_cloth.py:141-156) and the Duffing CSVs (benchmark_lqr_classic.py:174-178).
Pure host-side I/O; the arrays go straight into KoopmanNystromRegressor.fit.
"""
import os

import numpy as np


def load_cloth_experiment(path, n_trajs=50, n_inputs=6):
    """benchmark_lqr_cloth.py:148-152: trajectory i is `state_samples_cloth_swing_{i}.csv` (rows = time, 192
    interleaved x,y,z node coordinates) and `input_samples_cloth_swing_{i}.csv` (first `n_inputs` columns).
    Returns (trajs, controls): lists of (192 x T) and (n_inputs x T) arrays."""
    trajs, controls = [], []
    for i in range(n_trajs):
        tr = np.loadtxt(os.path.join(path, f"state_samples_cloth_swing_{i}.csv"), delimiter=",").T
        u = np.loadtxt(os.path.join(path, f"input_samples_cloth_swing_{i}.csv"), delimiter=",")[:, :n_inputs].T
        trajs.append(tr)
        controls.append(u)
    return trajs, controls


def cloth_splits(trajs, controls, n_val_trajs=10):
    """benchmark_lqr_cloth.py:153-156: the first `n_val_trajs` trajectories are the hyper-parameter validation set,
    the rest the system-identification pool."""
    return (trajs[:n_val_trajs], controls[:n_val_trajs]), (trajs[n_val_trajs:], controls[n_val_trajs:])


def load_duffing(path):
    """benchmark_lqr_classic.py:174-178: forced + unforced snapshot pairs; the unforced block gets zero input.
    Returns X ((2+1) x N) = [state; input] and Y (2 x N)."""
    xf = np.loadtxt(os.path.join(path, "duffing_x_forced.csv"), delimiter=",")
    xu = np.loadtxt(os.path.join(path, "duffing_x_unforced.csv"), delimiter=",")
    uf = np.loadtxt(os.path.join(path, "duffing_u_forced.csv"), delimiter=",").reshape(1, -1)
    yf = np.loadtxt(os.path.join(path, "duffing_y_forced.csv"), delimiter=",")
    yu = np.loadtxt(os.path.join(path, "duffing_y_unforced.csv"), delimiter=",")
    X = np.vstack((np.hstack((xf, xu)), np.hstack((uf, np.zeros((1, xu.shape[1]))))))
    Y = np.hstack((yf, yu))
    return X, Y


def cloth_reference_state(initial_state, alpha=np.pi / 4, vertical_shift=0.0, horizontal_shift=0.0):
    """The swing-up reference of benchmark_lqr_cloth.py:241-255: every node is rotated by `alpha` about the top edge
    (x unchanged, y += r sin(alpha), z += r - r cos(alpha) with r = |z_top - z_node|)."""
    x = np.asarray(initial_state, dtype=np.float64).reshape(-1, 1)
    n_states = x.shape[0]
    offset = np.zeros((n_states, 1))
    z_top = x[-1]
    for i in range(n_states):
        if i % 3 == 1:
            r = abs(z_top - x[i + 1])
            offset[i] = r * np.sin(alpha) + horizontal_shift
        if i % 3 == 2:
            r = abs(z_top - x[i])
            offset[i] = r - r * np.cos(alpha) + vertical_shift
    return x + offset
