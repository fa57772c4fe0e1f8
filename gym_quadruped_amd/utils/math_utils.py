"""Small math helpers - mirror of the reference's ``gym_quadruped/utils/math_utils.py`` (same names/semantics)."""
from __future__ import annotations

import numpy as np


def skew(x):
    """Skew-symmetric matrix of a 3-vector (reference math_utils.py:7)."""
    x = np.asarray(x, dtype=float)
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def homogenous_transform(vec: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Apply a 4x4 homogeneous transform to a 3-vector (reference math_utils.py:12)."""
    vec = np.asarray(vec, dtype=float).reshape(-1)
    assert vec.shape == (3,), f'Expected 3D vector, got {vec} of shape {vec.shape}'
    assert X.shape == (4, 4) and X[3, 3] == 1, f'Expected homogeneous transformation matrix, got {X}'
    return (X @ np.append(vec, 1.0))[:3]


def angle_between_vectors(vector1, vector2) -> float:
    """Heading of ``vector2 - vector1`` in the xy plane (reference math_utils.py:37): atan2(dy, dx)."""
    d = np.asarray(vector2, dtype=float) - np.asarray(vector1, dtype=float)
    return float(np.arctan2(d[1], d[0]))


def _process_range(values):
    """Scalar -> (v, v); 2-sequence -> itself (reference math_utils.py:54)."""
    if isinstance(values, (int, float, np.number)):
        return (values, values)
    if isinstance(values, (tuple, list, np.ndarray)):
        assert len(values) == 2, f'Invalid range values, expected (min, max) got: {values}'
        return values
    return None
