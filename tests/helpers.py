"""Shared helpers for the test-suite (deterministic fills, camera rigs)."""

import numpy as np


def formula_fill(shape, salt):
    """Same closed-form fill tests/golden/make_goldens.py used for full-size models."""
    n = int(np.prod(shape))
    fan_in = shape[-1] if len(shape) > 1 else shape[0]
    h = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(salt * 40503 + 12345))
    h = (h % np.uint64(2 ** 32)) >> np.uint64(8)
    vals = (h.astype(np.float64) / 2 ** 24 * 2 - 1) / np.sqrt(fan_in)
    return vals.astype(np.float32).reshape(shape)


def look_at_camera(eye, width, height, fov_deg=40.0):
    """Camera-to-world pose looking at the origin (x right, y down, z forward)."""
    eye = np.asarray(eye, np.float32)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0, 1, 0], np.float32)
    if abs(np.dot(fwd, up)) > 0.95:
        up = np.array([1, 0, 0], np.float32)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    ext = np.eye(4, dtype=np.float32)
    ext[:3, 0], ext[:3, 1], ext[:3, 2], ext[:3, 3] = right, down, fwd, eye
    focal = 0.5 * width / np.tan(0.5 * np.deg2rad(fov_deg))
    intr = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1]], np.float32)
    return intr, ext
