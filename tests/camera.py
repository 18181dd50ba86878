"""Test helper: camera matrices in glm's memory order (m[c] = column c), fp32 — plausible inputs for the prepass; nothing
here needs to match glm bit for bit because the matrices are INPUTS of the code under test."""
import numpy as np


def perspective(fov_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    """glm::perspective (RH, depth -1..1): renderer.cpp:189-191."""
    t = np.tan(np.radians(fov_deg) / 2.0)
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = -(far + near) / (far - near)
    m[2, 3] = -1.0
    m[3, 2] = -(2.0 * far * near) / (far - near)
    return m


def look_at(eye, center, up=(0, 1, 0)) -> np.ndarray:
    """glm::lookAt (RH)."""
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, 0], m[1, 0], m[2, 0] = s
    m[0, 1], m[1, 1], m[2, 1] = u
    m[0, 2], m[1, 2], m[2, 2] = -f
    m[3, 0], m[3, 1], m[3, 2] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m.astype(np.float32)


def trs(translate=(0, 0, 0), rot_axis=(0, 1, 0), rot_deg=0.0, scale=(1, 1, 1)) -> np.ndarray:
    a = np.asarray(rot_axis, np.float64)
    a /= np.linalg.norm(a)
    c, s = np.cos(np.radians(rot_deg)), np.sin(np.radians(rot_deg))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    Rm = np.eye(3) + s * K + (1 - c) * (K @ K)                    # math matrix (row-major)
    Mm = np.eye(4)
    Mm[:3, :3] = Rm @ np.diag(scale)
    Mm[:3, 3] = translate
    return Mm.T.astype(np.float32)                                # -> column-major memory order
