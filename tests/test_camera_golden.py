"""Camera/pose conventions against golden vectors produced by the reference's own
importable functions (tests/golden/make_golden.py -> pose_golden.json), and the oracle against its
committed config-1 fixture."""
import json
import math
import os

import numpy as np
import pytest

import oracle_c
import oracle_np as onp
from sage_gs import camera

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def poses():
    return json.load(open(os.path.join(GOLD, "pose_golden.json")))["cases"]


def _wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


def test_rotation_encoding_matches_reference_transform(poses):
    """trajectory_2d_to_3d.transform_trajectory_points: rotation = [-sin(psi/2),0,0,cos(psi/2)], psi = yaw+pi."""
    for c in poses:
        got = camera.rotation_from_yaw(c["yaw"])
        assert np.allclose(got, c["points_after"][0]["rotation"], atol=1e-12), c["yaw"]
        assert c["points_after"][-1]["rotation"] == [0.0, 0.0, 0.0, 1.0]      # the reference resets the last waypoint


def test_yaw_decoding_matches_action_generator(poses):
    """generate_actions.BatchActionGenerator.yaw_from_quaternion on the stored rotation."""
    for c in poses:
        rot = c["points_after"][0]["rotation"]
        assert abs(camera.yaw_from_rotation(rot) - c["action_generator_yaw"]) < 1e-12
        # and the environment's start heading (simple_env.py:1149-1182) undoes the +pi of the transform
        assert abs(_wrap(camera.env_start_yaw(rot) - c["yaw"])) < 1e-9


def test_view_matrix_of_datagen_pose(poses):
    """generate_images.py:417-421 passes the stored 4-vector as Isaac's (w,x,y,z): a rotation about +Z by
    psi + pi; the camera looks along R(q)(+X), eye height forced to 1.2 m."""
    for c in poses:
        pt = c["points_after"][0]
        pos, orient = camera.datagen_pose(pt)
        assert pos[2] == 1.2 and pos[:2] == pt["position"][:2]
        V = camera.view_from_isaac_pose(pos, orient)
        R = V[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        theta = c["action_generator_yaw"] + math.pi
        fwd = np.array([math.cos(theta), math.sin(theta), 0.0])
        assert np.allclose(R[2], fwd, atol=1e-9)                # camera +Z = heading
        assert np.allclose(R[1], [0, 0, -1], atol=1e-12)        # camera +Y = world down (Z-up stage)
        assert np.allclose(V @ np.array([*pos, 1.0]), [0, 0, 0, 1], atol=1e-12)
        # a point one metre ahead lands on the optical axis, one metre deep
        ahead = np.array(pos) + fwd
        assert np.allclose(V @ np.array([*ahead, 1.0]), [0, 0, 1, 1], atol=1e-9)


def test_env_orientation_restates_simple_env():
    """simple_env.py:1208-1256: +sin(-22.5deg) on component 0, yaw delta composed above 0.01 rad."""
    q = [0.3, 0.0, 0.0, 0.95]
    base = camera.env_orientation(q)
    assert np.allclose(base, [0.3 + math.sin(math.radians(-22.5)), 0.0, 0.0, 0.95])
    assert camera.env_orientation(q, 1.0, 1.005) == base
    d = 0.4
    got = camera.env_orientation(q, 1.4, 1.0)
    bx, bw = base[0], base[3]
    assert np.allclose(got, [bx * math.cos(d / 2) - bw * math.sin(d / 2), 0.0, 0.0, bw * math.cos(d / 2) + bx * math.sin(d / 2)])


def test_reference_intrinsics():
    fx, fy, cx, cy = camera.reference_intrinsics(640, 480)         # simple_env.py:52 default resolution
    assert abs(fx - 640 * 8.0 / 20.955) < 1e-9 and fx == fy and (cx, cy) == (320.0, 240.0)
    assert abs(math.degrees(2 * math.atan(320 / fx)) - 105.3) < 0.1   # HFOV of the reference lens


# ---- the oracle against its committed fixture ----------------------------------------------------------
@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "config1_golden.npz"))


@pytest.mark.parametrize("which", ["numpy", "c"])
def test_oracle_reproduces_config1_golden(gold, which):
    scene, _ = onp.config1_scene(n=int(gold["n"]), seed=int(gold["seed"]))
    cam = onp.Camera(int(gold["width"]), int(gold["height"]), float(gold["f"]), float(gold["f"]), 64.0, 64.0,
                     np.eye(4, dtype=np.float32))
    if which == "numpy":
        img, aux = onp.render(*scene, cam)
        rect, tiles, depth = aux["pre"]["rect"], aux["pre"]["tiles"], aux["pre"]["depth"].view(np.uint32)
        vis = aux["pre"]["visible"]
    else:
        img, aux = oracle_c.render(*scene, cam)
        rect, tiles, depth = aux["rect"], aux["tiles"], aux["depth_bits"]
        vis = tiles > 0
    assert aux["D"] == int(gold["D"]) and aux["D_f"] == int(gold["D_f"]) and aux["n_visible"] == int(gold["n_visible"])
    assert (aux["offsets"] == gold["offsets"]).all() and (aux["ids"] == gold["ids"]).all()
    assert (tiles == gold["tiles"]).all() and (rect[vis] == gold["rect"][vis]).all() and (depth == gold["depth_bits"]).all()
    assert (aux["n_contrib"] == gold["n_contrib"]).all()
    assert np.abs(np.asarray(img, np.float64) - gold["image"]).max() < 1e-6


# ---- the benchmark environment's pose branch, against the reference's own methods (pose_env_golden.json) ----
@pytest.fixture(scope="module")
def env_poses():
    return json.load(open(os.path.join(GOLD, "pose_env_golden.json")))


def test_env_start_pose_matches_simple_env(env_poses):
    """SimpleVLNEnv.set_start_pose (simple_env.py:1149-1195): the yaw it derives and the pose it hands to the camera."""
    for c in env_poses["cases"]:
        yaw0 = camera.env_start_yaw(c["rotation_xyzw"])
        assert abs(yaw0 - c["start_yaw"]) < 1e-12
        assert abs(_wrap(yaw0 - c["trajectory_yaw"])) < 1e-9            # it undoes the +pi of the trajectory transform
        pos, o = camera.env_pose(c["position"], c["rotation_xyzw"], yaw0, yaw0)
        assert pos.dtype == np.float32 and o.dtype == np.float32 and c["dtypes"] == ["float32", "float32"]
        assert pos.tolist() == c["start_camera_position"]              # float32 of the same numbers: bit for bit
        assert np.allclose(o, c["start_orientation"], rtol=0, atol=1e-7)


def test_env_camera_update_matches_simple_env(env_poses):
    """SimpleVLNEnv._update_camera_position (simple_env.py:1196-1284) after yaw changes on both sides of its 0.01 rad switch."""
    n_switch = 0
    for c in env_poses["cases"]:
        for s in c["steps"]:
            pos, o = camera.env_pose(s["agent_position"], c["rotation_xyzw"], s["yaw"], c["start_yaw"])
            assert pos.tolist() == s["camera_position"] and abs(pos[2] - 1.2) < 1e-6
            assert np.allclose(o, s["orientation"], rtol=0, atol=1e-7), (c["trajectory_yaw"], s["yaw"])
            n_switch += abs(s["yaw"] - c["start_yaw"]) > 0.01
    assert n_switch >= 3 * len(env_poses["cases"])                     # the composed branch was exercised
    for f in env_poses["fallback"]:
        pos, o = camera.env_pose([1.0, 2.0, 0.3], None, f["yaw"])
        assert pos.tolist() == f["camera_position"] and np.allclose(o, f["orientation"], rtol=0, atol=1e-7)


def test_env_pose_gives_a_rigid_level_view(env_poses):
    """The environment's 4-vector is not a unit quaternion (simple_env.py:1212-1221 adds a scalar to one component); Isaac
    Sim normalises it, and so does view_from_isaac_pose: the view stays rigid and the camera stays level (a rotation
    about +Z only), looking along the heading the 4-vector encodes when read scalar-first."""
    for c in env_poses["cases"]:
        for s in [dict(camera_position=c["start_camera_position"], orientation=c["start_orientation"])] + c["steps"]:
            V = camera.view_from_isaac_pose(s["camera_position"], s["orientation"])
            R = V[:3, :3]
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
            w, x, y, z = s["orientation"]                     # Isaac reads the 4-vector scalar-first
            assert abs(x) < 1e-12 and abs(y) < 1e-12
            a = 2.0 * math.atan2(z, w)                        # a rotation about +Z: the camera stays level
            assert np.allclose(R[2], [math.cos(a), math.sin(a), 0.0], atol=1e-9)      # camera +Z = heading
            assert np.allclose(R[1], [0.0, 0.0, -1.0], atol=1e-9)                     # camera +Y = world down


def test_isaac_pose_from_view_inverts_view_from_isaac_pose():
    from sage_gs import scenes
    rng = np.random.default_rng(8)
    for _ in range(50):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pos = rng.uniform(-5, 5, 3)
        V = camera.view_from_isaac_pose(pos, q)
        p2, q2 = camera.isaac_pose_from_view(V)
        assert np.allclose(p2, pos, atol=1e-12) and np.allclose(camera.view_from_isaac_pose(p2, q2), V, atol=1e-12)
    for yaw in (0.0, 1.0, math.pi, -2.5):                    # the pose lists of the benchmarks (level cameras)
        V = scenes.view_from_yaw((1.0, 2.0, 1.2), yaw)
        p2, q2 = camera.isaac_pose_from_view(V)
        assert np.allclose(camera.view_from_isaac_pose(p2, q2), V, atol=1e-12)


def test_pose_projection_and_asset_transform_validation():
    """Python surface (renderer._rigid / _check_model_to_world): a camera pose with fp32-composition noise is projected onto the nearest
    rotation, an orthonormal one passes bit for bit; a scene's model_to_world passes as it is when rigid to 2e-6, is projected when it is
    off by rounding noise (a product of fp32 rotations: up to the 1e-5 the library itself tolerates) and is REFUSED beyond that — a USD
    xformOp:scale of 1.0003 is a real scale (it used to be silently re-orthonormalised away)."""
    from sage_gs import renderer, scenes
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    V = np.eye(4); V[:3, :3] = q; V[:3, 3] = [0.3, -1.2, 2.0]
    assert np.array_equal(renderer._rigid(V), V)
    noisy = V.copy(); noisy[:3, :3] += 3e-5 * rng.normal(size=(3, 3))
    fixed = renderer._rigid(noisy)
    assert np.abs(fixed[:3, :3] @ fixed[:3, :3].T - np.eye(3)).max() < 1e-12 and np.abs(fixed - noisy).max() < 2e-4
    assert np.array_equal(fixed[:3, 3], noisy[:3, 3])
    far = V.copy(); far[:3, :3] *= 1.01
    assert np.array_equal(renderer._rigid(far), far)                  # not rounding noise: left for the library to refuse
    assert np.array_equal(renderer._check_model_to_world(scenes.MODEL_TO_WORLD), np.asarray(scenes.MODEL_TO_WORLD, np.float64))
    # a rotation composed in fp32 (two fp32 rotations multiplied in fp32): 1e-6 .. 5e-6 off — accepted, as the nearest rotation
    a32 = q.astype(np.float32); b32 = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    comp = np.eye(4); comp[:3, :3] = (a32 @ b32 @ a32.T @ b32).astype(np.float64) + 3e-6 * rng.normal(size=(3, 3))
    comp[:3, :3] *= np.sign(np.linalg.det(comp[:3, :3]))
    dev = np.abs(comp[:3, :3] @ comp[:3, :3].T - np.eye(3)).max()
    assert 2e-6 < dev < 1e-5, dev
    got = renderer._check_model_to_world(comp)
    assert np.abs(got[:3, :3] @ got[:3, :3].T - np.eye(3)).max() < 1e-12 and np.abs(got - comp).max() < 1e-5 and np.array_equal(got[:3, 3], comp[:3, 3])
    for bad in (np.diag([1.0003, 1.0003, 1.0003, 1.0]), np.diag([1.0, -1.0, 1.0, 1.0]), np.array([[1, 1e-3, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])):
        with pytest.raises(ValueError, match="not a rigid transform"):
            renderer._check_model_to_world(bad)
