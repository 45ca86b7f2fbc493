"""CPU: the data-format readers against files written in the reference's layout, and (when /root/reference is
mounted) against the reference's own files and the committed golden copy of them."""
import os

import numpy as np
import pytest

from nys_koop_lqr_amd import datasets, harness

REF = "/root/reference"


def test_cloth_and_duffing_layouts(tmp_path):
    rng = np.random.default_rng(0)
    for i in range(3):
        np.savetxt(tmp_path / f"state_samples_cloth_swing_{i}.csv", rng.standard_normal((7, 192)), delimiter=",")
        np.savetxt(tmp_path / f"input_samples_cloth_swing_{i}.csv", rng.standard_normal((7, 9)), delimiter=",")
    trajs, ctrls = datasets.load_cloth_experiment(str(tmp_path), n_trajs=3)
    assert [t.shape for t in trajs] == [(192, 7)] * 3 and [c.shape for c in ctrls] == [(6, 7)] * 3
    (vt, vc), (tt, tc) = datasets.cloth_splits(trajs, ctrls, n_val_trajs=1)
    assert len(vt) == 1 and len(tt) == 2
    X, Y = harness.create_data_matrices(tt, tc, range(2))
    assert X.shape == (198, 12) and Y.shape == (192, 12) and np.array_equal(Y[:, 0], tt[0][:, 1])
    for name, shape in (("x_forced", (2, 5)), ("x_unforced", (2, 4)), ("y_forced", (2, 5)), ("y_unforced", (2, 4))):
        np.savetxt(tmp_path / f"duffing_{name}.csv", rng.standard_normal(shape), delimiter=",")
    np.savetxt(tmp_path / "duffing_u_forced.csv", rng.standard_normal(5), delimiter=",")
    X, Y = datasets.load_duffing(str(tmp_path))
    assert X.shape == (3, 9) and Y.shape == (2, 9) and np.all(X[2, 5:] == 0)
    ref = datasets.cloth_reference_state(trajs[0][:, 0])
    assert ref.shape == (192, 1) and np.array_equal(ref[0::3], trajs[0][0::3, :1])  # x coordinates unchanged


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted")
def test_readers_on_the_reference_files(golden):
    trajs, ctrls = datasets.load_cloth_experiment(os.path.join(REF, "8x8_cloth_swing_xyz"), n_trajs=12)
    g = golden("f6_cloth_known_gain.npz")  # trajectories 10..39 as the reference's scripts slice them
    assert np.array_equal(trajs[10], g["trajs"][0]) and np.array_equal(ctrls[11], g["inputs"][1])
    X, Y = datasets.load_duffing(os.path.join(REF, "duffing"))
    assert X.shape == (3, 69900) and Y.shape == (2, 69900)


def test_cloth_reference_state_matches_shipped_csv(golden):
    """datasets.cloth_reference_state (benchmark_lqr_cloth.py:241-255) against the reference's shipped
    8x8_cloth_swing_xyz/sim_results/nystrom/data/reference_lqr.csv (kept in the f10 golden; 5-digit CSV)."""
    from nys_koop_lqr_amd import datasets
    g = golden("f10_lqr_control.npz")
    got = datasets.cloth_reference_state(g["initial_state"])
    assert got.shape == g["reference_lqr"].shape
    assert np.max(np.abs(got - g["reference_lqr"])) < 5e-6
