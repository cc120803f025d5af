"""'Next' rows of SURVEY.md §8f, host side (no GPU): PLY loader round trips, USDA path resolver, sweep
input parsing and camera list.  The GPU side (frames through the adapter / sweep) is in test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

import parity_cases as pc
from sage_gs import adapter, camera, ply, sweep


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_standard_ply_round_trip(tmp_path, deg):
    m, s, q, o, sh, d = pc.random_scene(777, 3 + deg, deg)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    path = str(tmp_path / "scene.ply")
    ply.save_ply(path, m, s, q, o, sh, d)
    m2, s2, q2, o2, sh2, d2 = ply.load_ply(path)
    assert d2 == deg and sh2.shape == sh.shape
    assert np.array_equal(m2, m) and np.array_equal(sh2, sh)                 # stored verbatim
    assert np.allclose(s2, s, rtol=2e-6) and np.allclose(o2, o, atol=2e-6) and np.allclose(q2, q, atol=1e-6)


def test_ply_property_order_and_layout(tmp_path):
    """f_rest is channel-major in the file (R coefficients 1..15, then G, then B) and rot_0 is w."""
    n, k = 5, 16
    props = ["opacity", "rot_3", "rot_0", "rot_1", "rot_2", "z", "y", "x", "scale_2", "scale_1", "scale_0",
             "f_dc_2", "f_dc_1", "f_dc_0"] + [f"f_rest_{i}" for i in reversed(range(45))]
    arr = np.zeros(n, dtype=[(p, "<f4") for p in props])
    arr["x"], arr["y"], arr["z"] = 1, 2, 3
    arr["rot_0"] = 2.0                                            # un-normalised w
    arr["opacity"] = 0.0
    arr["scale_0"], arr["scale_1"], arr["scale_2"] = np.log(0.5), np.log(0.25), 0.0
    for i in range(45):
        arr[f"f_rest_{i}"] = i
    for c in range(3):
        arr[f"f_dc_{c}"] = 10 + c
    path = str(tmp_path / "odd.ply")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
        for p in props:
            f.write(f"property float {p}\n".encode())
        f.write(b"end_header\n"); f.write(arr.tobytes())
    m, s, q, o, sh, d = ply.load_ply(path)
    assert d == 3 and np.allclose(m[0], [1, 2, 3]) and np.allclose(s[0], [0.5, 0.25, 1.0])
    assert np.allclose(q[0], [1, 0, 0, 0]) and np.allclose(o, 0.5)
    assert np.allclose(sh[0, 0], [10, 11, 12])
    assert sh[0, 1, 0] == 0 and sh[0, 15, 0] == 14 and sh[0, 1, 1] == 15 and sh[0, 1, 2] == 30 and sh[0, 15, 2] == 44


def test_compressed_ply_round_trip(tmp_path):
    """Encoder/decoder pair of the PlayCanvas layout agree to the quantisation step (experimental format)."""
    m, s, q, o, sh, _ = pc.random_scene(1000, 9, 0)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    path = str(tmp_path / "c.ply")
    ply.save_compressed_ply(path, m, s, q, o, sh, 0)
    m2, s2, q2, o2, sh2, d2 = ply.load_compressed_ply(path)
    assert d2 == 0
    span = m.max(0) - m.min(0)
    assert np.abs(m2 - m).max() < span.max() / 1000.0
    assert np.abs(np.log(s2) - np.log(s)).max() < 0.01
    assert np.abs(o2 - o).max() <= 0.5 / 255 + 1e-6
    dot = np.abs(np.sum(q2 * q, axis=1))                         # q and -q are the same rotation
    assert dot.min() > 0.9999
    col = np.clip(0.5 + ply.SH_C0 * sh[:, 0], 0, 1); col2 = 0.5 + ply.SH_C0 * sh2[:, 0]
    assert np.abs(col2 - col).max() <= 0.5 / 255 + 1e-6


def test_usda_resolver_on_the_reference_template_shape():
    text = '''
    def Xform "World" {
        over "gauss" (
            prepend references = @/data/scenes/0042.usdz[gauss.usda]@
        ) {
            float3 xformOp:rotateXYZ = (-90, 0, 0)
        }
        def "scene_collision" ( prepend payload = @/data/collision/0042_collision.usd@ ) {}
    }'''
    got = adapter.parse_scene_usda(text)
    assert got == {"usdz": "/data/scenes/0042.usdz", "collision": "/data/collision/0042_collision.usd",
                   "rotate_xyz": (-90.0, 0.0, 0.0)}


def test_sweep_input_and_cameras(tmp_path):
    gt = {"groundtruth_data": [
        {"trajectory_id": "7", "instruction_index": 0, "sampled_points": [
            {"point_id": 0, "position": [1.0, 2.0, 0.0], "rotation": camera.rotation_from_yaw(0.3)},
            {"point_id": 1, "position": [1.5, 2.0, 0.0], "rotation": camera.rotation_from_yaw(0.4)}]},
        {"trajectory_id": "7", "instruction_index": 1, "sampled_points": []},        # same trajectory, 2nd instruction
        {"trajectory_id": "9", "instruction_index": 0, "sampled_points": [
            {"point_id": 0, "position": [0.0, 0.0, 0.5], "rotation": [0.0, 0.0, 0.0, 1.0]}]}]}
    p = tmp_path / "action_groundtruth.json"
    p.write_text(json.dumps(gt))
    trs = sweep.load_trajectories(str(p))
    assert [t["trajectory_id"] for t in trs] == ["7", "9"] and len(trs[0]["points"]) == 2
    cams = sweep.cameras_for(trs[0]["points"])
    assert (cams[0].width, cams[0].height) == (1024, 768) and abs(cams[0].fx - 1024 * 8 / 20.955) < 1e-9
    eye = -cams[0].view[:3, :3].T @ cams[0].view[:3, 3]
    assert np.allclose(eye, [1.0, 2.0, 1.2])                      # eye height forced to 1.2 m
