"""Camera conventions of the reference's call sites, restated for the renderer.

The reference never builds a view matrix: it hands `(position, orientation)` to Isaac Sim's
`Camera.set_world_pose` and lets the simulator do the rest.  These helpers reproduce the arithmetic the
reference applies on ITS side of that call, and define — this repo's own definition, since Isaac Sim is
closed — the view matrix that results:

  * Isaac Sim orientations are scalar-first (w, x, y, z) and, with the default "world" camera axes, an
    identity orientation looks along +X with +Z up.  The renderer's camera is +Z forward, +X right,
    +Y down, so   forward = R(q)·(+X),  up = R(q)·(+Z),  rows of the view rotation = (forward x up, -up, forward).
  * trajectory files store `rotation = [qx, qy, qz, qw] = [-sin(psi/2), 0, 0, cos(psi/2)]`, psi = yaw + pi
    (`trajectory_2d_to_3d.py:154-171`); `generate_actions.py:129-133` decodes it as yaw = 2·atan2(-qx, qw).
  * the data-generation loop passes that 4-vector UNCHANGED as the Isaac orientation and forces the eye
    height to 1.2 m (`generate_images.py:417-421`); read scalar-first it is a rotation about +Z by psi + pi.
  * the benchmark environment adds sin(-45 deg / 2) to component 0 and optionally composes a yaw delta
    (`simple_env.py:1196-1238`), un-normalised; Isaac Sim normalises quaternions, so do we.

Golden vectors come from running the reference's own code (`tests/golden/make_golden.py`): the importable pose
functions -> `tests/golden/pose_golden.json`, and `SimpleVLNEnv.set_start_pose` / `_update_camera_position` themselves,
on an instance made without the simulator -> `tests/golden/pose_env_golden.json`.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

EYE_HEIGHT = 1.2                         # generate_images.py:45, simple_env.py:1204
REF_FOCAL_OVER_APERTURE = 8.0 / 20.955   # focalLength 8.0 (simple_env.py:905) on UsdGeom.Camera's default aperture


def rotation_from_yaw(yaw: float):
    """Trajectory encoding of a heading (trajectory_2d_to_3d.py:154-171): [qx, qy, qz, qw]."""
    psi = yaw + math.pi
    if psi > math.pi:
        psi -= 2 * math.pi
    return [-math.sin(psi / 2.0), 0.0, 0.0, math.cos(psi / 2.0)]


def yaw_from_rotation(rotation: Sequence[float]) -> float:
    """Inverse used by the action generator (generate_actions.py:129-133): the heading psi the file encodes."""
    qx, _, _, qw = rotation
    return 2.0 * math.atan2(-qx, qw)


def quat_to_matrix_wxyz(q: Sequence[float]) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    if n == 0.0:
        raise ValueError("zero quaternion")
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def view_from_isaac_pose(position: Sequence[float], orientation_wxyz: Sequence[float]) -> np.ndarray:
    """world->camera 4x4 for `cam.set_world_pose(position, orientation)` (simple_env.py:1284;
    generate_images.py:419-421), camera axes "world": +X forward, +Z up at identity."""
    # (scalar arithmetic: this runs once per get_rgba() of the adapter, and the small-array NumPy form of the same eight lines took 50 us of a
    #  0.5-ms call)
    w, x, y, z = (float(v) for v in orientation_wxyz)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    if n == 0.0:
        raise ValueError("zero quaternion")
    w, x, y, z = w / n, x / n, y / n, z / n
    fx, fy, fz = 1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)          # the rotation's first column: forward
    ux, uy, uz = 2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)          # its third column: up
    rx, ry, rz = fy * uz - fz * uy, fz * ux - fx * uz, fx * uy - fy * ux                    # right = forward x up
    px, py, pz = (float(v) for v in position)
    return np.array([[rx, ry, rz, -(rx * px + ry * py + rz * pz)],
                     [-ux, -uy, -uz, ux * px + uy * py + uz * pz],
                     [fx, fy, fz, -(fx * px + fy * py + fz * pz)],
                     [0.0, 0.0, 0.0, 1.0]])


def isaac_pose_from_view(view) -> tuple:
    """Inverse of view_from_isaac_pose: (position [3], orientation (w,x,y,z)) of the `cam.set_world_pose` call that gives this
    world->camera matrix (rows right, down, forward).  Lets pose lists kept as view matrices (scenes.room_cameras) drive a GsCamera."""
    V = np.asarray(view, np.float64).reshape(4, 4)
    rot = V[:3, :3]
    pos = -rot.T @ V[:3, 3]
    fwd, up = rot[2], -rot[1]
    left = np.cross(up, fwd)
    R = np.stack([fwd, left, up], 1)                        # columns: the camera's +X (forward), +Y (left), +Z (up) in the world
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] >= R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    return pos, np.asarray(q, np.float64)


def datagen_pose(point: dict):
    """(position, orientation) exactly as generate_images.py:417-421 passes them to the camera."""
    pos = [float(v) for v in point["position"]]
    pos[2] = EYE_HEIGHT
    return pos, [float(v) for v in point["rotation"]]


def env_orientation(original_quaternion: Sequence[float], yaw: float = None, initial_yaw: float = None):
    """The 4-vector simple_env.py:1208-1256 hands to set_world_pose: the trajectory quaternion with
    sin(-45deg/2) added to component 0, composed with the yaw change when it exceeds 0.01 rad."""
    qx_o, qy_o, qz_o, qw_o = (float(v) for v in original_quaternion)
    bx, by, bz, bw = qx_o + math.sin(math.radians(-45) / 2), qy_o, qz_o, qw_o
    if yaw is not None and initial_yaw is not None and abs(yaw - initial_yaw) > 0.01:
        d = yaw - initial_yaw
        qz_d, qw_d = math.sin(d / 2.0), math.cos(d / 2.0)
        return [bx * qw_d + bw * (-qz_d), by * qw_d, bz * qw_d, bw * qw_d - bx * (-qz_d)]
    return [bx, by, bz, bw]


def env_fallback_orientation(yaw: float):
    """simple_env.py:1258-1266: the orientation when no trajectory quaternion is known (before set_start_pose)."""
    return [-math.sin(yaw / 2.0), 0.0, 0.0, math.cos(yaw / 2.0)]


def env_pose(agent_position: Sequence[float], original_quaternion: Sequence[float] = None, yaw: float = None,
             initial_yaw: float = None):
    """(position, orientation) exactly as simple_env.py:1196-1284 passes them to `cam.set_world_pose`: the agent's xy at
    eye height 1.2 m, both as float32."""
    pos = np.asarray(agent_position, np.float32).copy()
    pos[2] = EYE_HEIGHT
    o = env_orientation(original_quaternion, yaw, initial_yaw) if original_quaternion is not None else env_fallback_orientation(yaw)
    return pos, np.asarray(o, np.float32)


def env_start_yaw(rotation_xyzw: Sequence[float]) -> float:
    """simple_env.py:1149-1182: the agent heading recovered from a trajectory quaternion."""
    x, _, _, w = rotation_xyzw
    yaw = 2 * math.atan2(-x, w) - math.pi
    if yaw < -math.pi:
        yaw += 2 * math.pi
    elif yaw > math.pi:
        yaw -= 2 * math.pi
    return yaw


def reference_intrinsics(width: int, height: int):
    """(fx, fy, cx, cy): 8 mm lens on the 20.955 mm USD aperture, square pixels, centred (A2)."""
    f = width * REF_FOCAL_OVER_APERTURE
    return f, f, width / 2.0, height / 2.0


def reference_camera(width: int, height: int, position, orientation_wxyz):
    """A renderer Camera for a reference pose: resolution + lens of the reference, pose as Isaac takes it."""
    from .renderer import Camera
    fx, fy, cx, cy = reference_intrinsics(width, height)
    return Camera(width, height, fx, fy, cx, cy, view_from_isaac_pose(position, orientation_wxyz))
