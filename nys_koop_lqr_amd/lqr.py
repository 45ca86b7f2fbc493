"""Host-side discrete LQR (the Riccati/DARE step stays on the host, BASELINE north_star).

Stands in for `control.dlqr(A, B, Q, R)` used at benchmark_lqr_cloth.py:262, benchmark_lqr_classic.py:288 and
benchmark_lqr_hjb.py:293,356 (python-control is not a dependency here): K = (B'PB + R)^-1 B'PA with P the
stabilising solution of the discrete algebraic Riccati equation.
"""
import numpy as np
import scipy.linalg


def dlqr(A, B, Q, R):
    """Returns (K, P).  control.dlqr returns (K, S, E); K and S(=P) are the same quantities."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(B.T @ P @ B + R, B.T @ P @ A)
    return K, P


def cloth_gain_for_simulator(K):
    """Row permutation expected by the MATLAB cloth simulator (benchmark_lqr_cloth.py:263)."""
    return K[[0, 3, 1, 4, 2, 5], :]
