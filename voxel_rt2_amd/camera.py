"""Host-side camera matrices.

The reference obtains its matrices from ``ti.ui.Camera`` (scene.py:188-191, 233-237), i.e.
``glm::perspective(radians(fov), aspect, near, far)`` and ``glm::lookAt`` handed over in glm
(column-major) memory order, and transposes them on load (pathtracer.py:266-268, 278-280).
Here the same matrices are built with numpy; the device receives row-major mathematical
matrices plus their inverses (inverted in float64, rounded to float32 once).
"""
import numpy as np


def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.sqrt(np.sum(v * v))


def perspective(fov_y_rad, aspect, z_near, z_far):
    """glm::perspective, right-handed, clip depth -1..1, as a mathematical (row, col) matrix."""
    f = 1.0 / np.tan(fov_y_rad / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = -(z_far + z_near) / (z_far - z_near)
    m[2, 3] = -(2.0 * z_far * z_near) / (z_far - z_near)
    m[3, 2] = -1.0
    return m


def look_at(eye, center, up):
    """glm::lookAt (right-handed) as a mathematical (row, col) matrix."""
    eye = np.asarray(eye, dtype=np.float64)
    f = _normalize(np.asarray(center, dtype=np.float64) - eye)
    s = _normalize(np.cross(f, np.asarray(up, dtype=np.float64)))
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return m


def to_glm_memory(m):
    """What ti.ui.Camera.get_*_matrix returns: float32, element [col, row]."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T)


def from_glm_memory(M):
    """pathtracer.py:266-268: mat[j, i] = M[i, j]."""
    return np.ascontiguousarray(np.asarray(M, dtype=np.float32).T)


def inverse_f32(m32):
    return np.linalg.inv(np.asarray(m32, dtype=np.float64)).astype(np.float32)


# the reference's fixed initial pose: scene.py:28-29, 188-191, pathtracer.py:89
DEFAULT_POS = (0.4, 0.5, 2.0)
DEFAULT_LOOK_AT = (0.0, 0.0, 0.0)
DEFAULT_UP = (0.0, 1.0, 0.0)
DEFAULT_FOV = float(np.deg2rad(50.0))
Z_NEAR, Z_FAR = 0.01, 10.0


def default_matrices(width, height, pos=DEFAULT_POS, look=DEFAULT_LOOK_AT, fov=DEFAULT_FOV):
    """(view, proj) float32 mathematical matrices for the reference's camera contract."""
    proj = perspective(np.deg2rad(np.rad2deg(fov)), width / height, Z_NEAR, Z_FAR).astype(np.float32)
    view = look_at(pos, look, DEFAULT_UP).astype(np.float32)
    return view, proj
