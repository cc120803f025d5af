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
    m2, s2, q2, o2, sh2, d2 = ply.load_compressed_ply(path, sh_decode="bin_centre")
    assert d2 == 0
    span = m.max(0) - m.min(0)
    assert np.abs(m2 - m).max() < span.max() / 1000.0
    assert np.abs(np.log(s2) - np.log(s)).max() < 0.01
    assert np.abs(o2 - o).max() <= 0.5 / 255 + 1e-6
    dot = np.abs(np.sum(q2 * q, axis=1))                         # q and -q are the same rotation
    assert dot.min() > 0.9999
    col = np.clip(0.5 + ply.SH_C0 * sh[:, 0], 0, 1); col2 = 0.5 + ply.SH_C0 * sh2[:, 0]
    assert np.abs(col2 - col).max() <= 0.5 / 255 + 1e-6


def _serialise_prim(spec, indent="    "):
    """A prim stanza from (type, name, value) triples and composition arcs — this test's own writer."""
    arcs = "".join(f"{indent}    {a} = {v}\n" for a, v in spec["arcs"])
    attrs = "".join(f"{indent}    {t} {n} = {v}\n" for t, n, v in spec["attrs"])
    return f"{indent}{spec['specifier']} (\n{arcs}{indent})\n{indent}{{\n{attrs}{indent}}}\n\n"


def test_usda_resolver_on_the_reference_template_shape():
    """f-3: `parse_scene_usda` on the stanzas a real scene stage holds.  tests/golden/usda_golden.json is the output of the
    reference's OWN builder (sage3d_usda_builder.build_usda_content on Data/template.usda, run by make_golden.py) reduced
    to data: the attributes of /World/gauss and /World/scene_collision as (type, name, value) triples — `double3
    xformOp:rotateXYZ`, `prepend references = @...usdz[gauss.usda]@`, `prepend payload = @..._collision.usd@`.  The stage
    text is re-serialised here and parsed by the product."""
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "usda_golden.json")))
    text = ('#usda 1.0\n(\n    customLayerData = {\n        dictionary omni_layer = {\n'
            f'            string authoring_layer = "{g["authoring_layer"]}"\n        }}\n    }}\n'
            f'    defaultPrim = "World"\n    metersPerUnit = {g["stage"]["metersPerUnit"]:g}\n    upAxis = "{g["stage"]["upAxis"]}"\n)\n\n'
            'def Xform "World"\n{\n' + _serialise_prim(g["gauss"]) +
            '    def Camera "MySensorCamera"\n    {\n        float focalLength = 12\n        double3 xformOp:scale = (2, 2, 2)\n    }\n\n' +
            _serialise_prim(g["scene_collision"]) + '}\n')
    got = adapter.parse_scene_usda(text)
    sid = g["scene_id"]
    assert got["usdz"] == g["usdz_path_template"].format(scene_id=sid).strip("@").replace("[gauss.usda]", "")
    assert got["collision"] == g["collision_path_template"].format(scene_id=sid).strip("@")
    assert g["authoring_layer"] == f"./{sid}.usda"
    want = {n: v for _, n, v in g["gauss"]["attrs"]}
    vec = lambda s_: tuple(float(v) for v in s_.strip("()").split(","))
    assert got["rotate_xyz"] == vec(want["xformOp:rotateXYZ"]) == (-90.0, 0.0, 0.0)
    assert got["scale"] == vec(want["xformOp:scale"]) and got["translate"] == vec(want["xformOp:translate"])   # the gauss prim's, not the camera's
    assert list(got["xform_op_order"]) == json.loads(want["xformOpOrder"])
    assert got["up_axis"] == g["stage"]["upAxis"] == "Z" and got["meters_per_unit"] == g["stage"]["metersPerUnit"] == 1.0
    # the ops compose to the asset transform the renderer folds into the view (template.usda:119-123)
    from sage_gs import scenes
    assert np.abs(adapter.asset_model_to_world(got) - scenes.MODEL_TO_WORLD).max() < 1e-15
    # a scaled asset is refused, not silently mis-rendered; other rotations compose as Rz Ry Rx
    with pytest.raises(ValueError, match="rigid"):
        adapter.asset_model_to_world(dict(got, scale=(2.0, 2.0, 2.0)))
    M = adapter.asset_model_to_world(dict(got, rotate_xyz=(0.0, 0.0, 90.0), translate=(1.0, 2.0, 3.0)))
    assert np.allclose(M @ [1, 0, 0, 1], [1, 3, 3, 1])


def test_compressed_ply_decoder_on_hand_assembled_bytes(tmp_path):
    """f-1: `load_compressed_ply` against a file assembled BY HAND from the published PlayCanvas layout — chunk rows of
    min/max floats, vertex rows of four packed uint32 words (position 11-10-11, rotation 2-10-10-10 with the index of
    the dropped largest component, log-scale 11-10-11, colour 8-8-8-8), an `sh` element of bytes — independent of this
    repo's encoder.  Expected values are worked out here from the bit fields."""
    import struct
    # one chunk: position box [0,10] x [-1,1] x [2,4]; log-scale box [-4,-2] x [-3,-3] x [0,1]; colour box [0,1] (rgb) explicit
    chunk = [0.0, -1.0, 2.0, 10.0, 1.0, 4.0, -4.0, -3.0, 0.0, -2.0, -3.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0]
    chunk_names = ["min_x", "min_y", "min_z", "max_x", "max_y", "max_z", "min_scale_x", "min_scale_y", "min_scale_z",
                   "max_scale_x", "max_scale_y", "max_scale_z", "min_r", "min_g", "min_b", "max_r", "max_g", "max_b"]
    pos = lambda x, y, z: (x << 21) | (y << 11) | z               # 11 | 10 | 11 bits
    rot = lambda largest, a, b, c: (largest << 30) | (a << 20) | (b << 10) | c
    col = lambda r, g, b, a: (r << 24) | (g << 16) | (b << 8) | a
    h = 512                                                       # 10-bit code of ~0: (512/1023 - 0.5) sqrt2 = 6.9e-4
    verts = [
        # x = max, y = min, z = mid-code 1024/2047;   identity-ish: largest = w (index 0), others ~0;   scale codes 0 / any / 2047;  red, opaque
        (pos(2047, 0, 1024), rot(0, h, h, h), pos(0, 777, 2047), col(255, 0, 0, 255)),
        # x = min, y = max, z = min;                  largest = x (index 1): (w,y,z) = (a,b,c) with a = 1023 -> +0.7071;     grey 128, alpha 51 = 0.2
        (pos(0, 1023, 0), rot(1, 1023, h, h), pos(2047, 0, 0), col(128, 128, 128, 51)),
        # largest = z (index 3): stored (w,x,y) = (0, 1023, h) -> w = -0.7071, x = +0.7071
        (pos(1, 1, 1), rot(3, 0, 1023, h), pos(1024, 512, 1024), col(0, 255, 64, 0)),
    ]
    sh_bytes = [[(7 * i + 40 * v) % 256 for i in range(45)] for v in range(3)]          # degree 3: 45 bytes per vertex
    path = str(tmp_path / "hand.compressed.ply")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement chunk 1\n")
        for n_ in chunk_names:
            f.write(f"property float {n_}\n".encode())
        f.write(b"element vertex 3\nproperty uint packed_position\nproperty uint packed_rotation\nproperty uint packed_scale\nproperty uint packed_color\n")
        f.write(b"element sh 3\n")
        for i in range(45):
            f.write(f"property uchar f_rest_{i}\n".encode())
        f.write(b"end_header\n")
        f.write(struct.pack("<18f", *chunk))
        for v in verts:
            f.write(struct.pack("<4I", *v))
        for row in sh_bytes:
            f.write(bytes(row))
    m, s, q, o, sh, deg = ply.load_compressed_ply(path, sh_decode="bin_centre")
    assert deg == 3 and sh.shape == (3, 16, 3) and m.dtype == np.float32
    # positions: lerp(min, max, code / (2^bits - 1))
    assert np.allclose(m[0], [10.0, -1.0, 2.0 + 2.0 * 1024 / 2047], atol=1e-6)
    assert np.allclose(m[1], [0.0, 1.0, 2.0], atol=1e-6)
    assert np.allclose(m[2], [10.0 / 2047, -1.0 + 2.0 / 1023, 2.0 + 2.0 / 2047], atol=1e-6)
    # scales: exp(lerp(min_scale, max_scale, code))
    assert np.allclose(s[0], np.exp([-4.0, -3.0, 1.0]), rtol=1e-6)
    assert np.allclose(s[1], np.exp([-2.0, -3.0, 0.0]), rtol=1e-6)
    assert np.allclose(s[2], np.exp([-4.0 + 2.0 * 1024 / 2047, -3.0, 1024 / 2047]), rtol=1e-6)
    # rotations (w, x, y, z): three components stored as (code/1023 - 0.5) sqrt2, the dropped one = sqrt(1 - sum^2) >= 0
    e = (h / 1023 - 0.5) * np.sqrt(2.0)
    big = np.sqrt(1 - 3 * e * e)
    assert np.allclose(q[0], [big, e, e, e], atol=1e-6)
    r2 = np.sqrt(0.5)
    assert np.allclose(q[1], [r2, np.sqrt(1 - 0.5 - 2 * e * e), e, e], atol=1e-6)               # largest = x
    assert np.allclose(q[2], [-r2, r2, e, np.sqrt(1 - 1.0 - e * e if 1 - 1.0 - e * e > 0 else 0.0)], atol=1e-3)   # largest = z (~0 here)
    assert np.allclose(np.linalg.norm(q[:2], axis=1), 1.0, atol=1e-6)
    # colour -> SH DC = (lerp(min, max, code/255) - 0.5) / C0; alpha -> opacity (already activated)
    C0 = 0.28209479177387814
    assert np.allclose(sh[0, 0], [(1.0 - 0.5) / C0, (0.0 - 0.5) / C0, (0.0 - 0.5) / C0], atol=1e-5)
    assert np.allclose(sh[1, 0], [(128 / 255 - 0.5) / C0] * 3, atol=1e-5)
    assert np.allclose(o, [1.0, 51 / 255, 0.0], atol=1e-7)
    # higher SH bands: bytes, channel-major in the file (15 R, 15 G, 15 B), value = ((byte + 0.5) / 256 - 0.5) * 8
    for v in range(3):
        for k in range(15):
            for c in range(3):
                assert abs(sh[v, 1 + k, c] - (((sh_bytes[v][c * 15 + k] + 0.5) / 256.0 - 0.5) * 8.0)) < 1e-6


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


class _FakeScene:
    def __init__(self, g):
        self.g, self.freed = g, False

    def free(self):
        self.freed = True


class _FakeRenderer:
    """Stands where sage_gs.Renderer stands behind isaac_shim / GsCamera in this CPU test: records what it is asked to draw and
    hands back a frame that encodes the camera (so the test can tell which pose each saved image came from)."""
    device = "cpu"

    def __init__(self):
        self.uploads, self.frames = [], []

    def upload(self, g):
        self.uploads.append(g)
        return _FakeScene(g)

    def render_rgba8_host(self, cam, scene, *, config=None, tonemap=None):
        self.frames.append((cam, scene))
        img = np.zeros((cam.height, cam.width, 4), np.uint8)
        img[..., 0] = len(self.frames) % 256; img[..., 3] = 255
        return img


def test_isaac_shim_runs_the_reference_frame_loop_unchanged(tmp_path, monkeypatch):
    """The calls the reference's frame generator makes on Isaac Sim — RECORDED by running generate_images.SequentialFastImageGenerator.
    process_single_file itself behind recording stand-ins (tests/golden/make_golden.py isaac_trace_fixture -> isaac_call_trace.json: 47
    calls for a four-waypoint trajectory) — replayed one by one against sage_gs.isaac_shim with a fake renderer: the stage's .usda is
    parsed, the Gaussians beside the referenced USDZ are loaded with the prim's model->world transform and uploaded ONCE, every frame is
    drawn from the pose the loop set (z forced to 1.2, the stored rotation as the Isaac orientation), with the lens the loop set."""
    import sys
    from sage_gs import isaac_shim, scenes
    trace = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "isaac_call_trace.json")))
    K = trace["constants"]
    # a scene stage as sage3d_usda_builder writes it (tests/golden/usda_golden.json holds the shape), its USDZ, the .ply beside it
    scene_dir = tmp_path / "InteriorGS_usdz"; scene_dir.mkdir()
    (scene_dir / "0042.usdz").write_bytes(b"")
    rng = np.random.default_rng(3)
    n = 50
    arrays = (rng.normal(size=(n, 3)).astype(np.float32), np.full((n, 3), 0.05, np.float32),
              np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)), np.full(n, 0.5, np.float32), rng.normal(size=(n, 1, 3)).astype(np.float32), 0)
    ply.save_ply(str(scene_dir / "0042.ply"), *arrays)
    usda = tmp_path / "usda" / "0042.usda"; usda.parent.mkdir()
    usda.write_text('#usda 1.0\n(\n    metersPerUnit = 1\n    upAxis = "Z"\n)\ndef Xform "World"\n{\n    over "gauss" (\n'
                    '        prepend references = @../InteriorGS_usdz/0042.usdz[gauss.usda]@\n    )\n    {\n'
                    '        double3 xformOp:rotateXYZ = (-90, 0, 0)\n        double3 xformOp:scale = (1, 1, 1)\n'
                    '        double3 xformOp:translate = (0, 0, 0)\n'
                    '        uniform token[] xformOpOrder = ["xformOp:translate", "xformOp:rotateXYZ", "xformOp:scale"]\n    }\n}\n')
    fake = _FakeRenderer()
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split(".")[0] in ("omni", "pxr", "isaacsim")}
    isaac_shim.configure(renderer=fake)
    isaac_shim._state["loader"] = None
    images, results, world = [], [], None
    try:
        assert isaac_shim.install(force=True)
        import importlib
        # the names the reference imports (the recorder registered its stand-ins under exactly these module attributes)
        roots = {"SimulationApp": importlib.import_module("omni.isaac.kit").SimulationApp, "omni.usd": importlib.import_module("omni.usd"),
                 "World": importlib.import_module("omni.isaac.core").World, "open_stage": importlib.import_module("omni.isaac.core.utils.stage").open_stage,
                 "Camera": importlib.import_module("omni.isaac.sensor").Camera}
        pxr = importlib.import_module("pxr")
        roots.update(Gf=pxr.Gf, UsdGeom=pxr.UsdGeom, UsdLux=pxr.UsdLux)
        assert not roots["open_stage"](usd_path=str(tmp_path / "missing.usda"))          # the reference's failure mode: False
        store = dict(roots)

        def lookup(name):
            if name in store:
                return store[name]
            head, _, attr = name.rpartition(".")
            assert head, f"the trace names {name!r}, which nothing returned"
            return getattr(lookup(head), attr)

        def decode(v):
            if isinstance(v, dict) and "ndarray" in v:
                return np.array(v["ndarray"], dtype=v["dtype"])
            if isinstance(v, dict) and "ref" in v:
                return lookup(v["ref"])
            if isinstance(v, list):
                return [decode(x) for x in v]
            return v
        for ev in trace["events"]:
            args = [decode(a) for a in ev["args"]]
            kwargs = {k: decode(v) for k, v in ev["kwargs"].items()}
            if "usd_path" in kwargs:
                kwargs["usd_path"] = str(tmp_path / kwargs["usd_path"])
            if "resolution" in kwargs:
                kwargs["resolution"] = tuple(kwargs["resolution"])
            res = lookup(ev["call"])(*args, **kwargs)
            store[ev["call"] + "()"] = res
            results.append((ev["call"], res))
            if ev["call"].endswith(".GetPrimAtPath") and res:
                store["prim"] = res
            if ev["call"] == "open_stage":
                assert res, "the recorded run opened its stage"
            if ev["call"] == "World":
                world = res
            if ev["call"] == "Camera().get_rgba":
                assert res is not None and res.size > 0
                images.append(res[:, :, :3].copy())
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("omni", "pxr", "isaacsim")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
        isaac_shim._state.update(renderer=None, stage=None, app=None)
    points = trace["sampled_points"]
    CAMERA_RESOLUTION, CAMERA_HEIGHT = tuple(K["CAMERA_RESOLUTION"]), K["CAMERA_HEIGHT"]
    WORLD_STEP_COUNT, RENDER_STEP_COUNT = K["WORLD_STEP_COUNT"], K["RENDER_STEP_COUNT"]
    assert CAMERA_RESOLUTION == (1024, 768) and K["CAMERA_FOCAL_LENGTH"] == 8.0 and CAMERA_HEIGHT == 1.2
    # one upload of the stage's Gaussians, with the prim's transform; freed when the stage closed
    assert len(fake.uploads) == 1
    g = fake.uploads[0]
    assert len(g) == n and np.allclose(g.means.numpy(), arrays[0]) and g.sh_degree == 0
    assert np.allclose(g.model_to_world, scenes.MODEL_TO_WORLD)
    # every frame: the loop's pose, the loop's lens, the stage's scene
    assert len(fake.frames) == len(points) and world.steps == WORLD_STEP_COUNT + RENDER_STEP_COUNT * len(points)
    for i, ((c, sc), point) in enumerate(zip(fake.frames, points)):
        assert (c.width, c.height) == CAMERA_RESOLUTION and abs(c.fx - 1024 * 8.0 / 20.955) < 1e-3 and c.fx == c.fy
        pos = np.array(point["position"], np.float32); pos[2] = CAMERA_HEIGHT
        assert np.allclose(c.view, camera.view_from_isaac_pose(pos, np.array(point["rotation"], np.float32)))
        assert sc.g is g and images[i].shape == (768, 1024, 3) and images[i][0, 0, 0] == (i + 1) % 256


def _kat_payload():
    """The known-answer vectors of tests/golden/compressed_ply_kat.json as ONE compressed scene: vertex i carries case i's word in its
    field (zeros elsewhere).  Returns (cases, chunk table [1,18], packed [n,4], sh bytes [n,45])."""
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "compressed_ply_kat.json")))
    ck = kat["chunk"]
    chunks = np.array([ck["min"] + ck["max"] + ck["min_scale"] + ck["max_scale"] + ck["min_rgb"] + ck["max_rgb"]], np.float32)
    cases = kat["cases"]
    n = len(cases)
    packed = np.zeros((n, 4), np.uint32)
    shb = np.zeros((n, 45), np.uint8)
    col = {"packed_position": 0, "packed_rotation": 1, "packed_scale": 2, "packed_color": 3}
    for i, c in enumerate(cases):
        if c["field"] == "sh_byte":
            shb[i, :] = c["byte"]
        elif c["field"] == "sh_order":
            shb[i, :] = c["bytes"]
        else:
            packed[i, col[c["field"]]] = int(c["word"], 16)
    return cases, chunks, packed, shb


def _check_kat(cases, means, scales, quats, opac, sh_dc, sh_rest, rtol=3e-6, mode="bin_centre"):
    """sh_rest: [n, 15, 3] ([coefficient][channel]) or its [n, 45] flattening, decoded with `mode`; any argument may be None (not checked)."""
    close = lambda a, b: np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=2e-6)
    key = {"bin_centre": "expect", "linear255": "expect_linear255", "bin_centre_ends": "expect_bin_centre_ends"}[mode]
    for i, c in enumerate(cases):
        f = c["field"]
        if means is None and f.startswith("packed"):
            continue
        if f == "sh_order" and sh_rest is not None and mode == "bin_centre":
            assert close(np.asarray(sh_rest[i]).reshape(15, 3), c["expect"]), (i, np.asarray(sh_rest[i]).reshape(15, 3)[:2])
        elif f == "sh_byte" and sh_rest is not None:
            got = np.asarray(sh_rest[i], np.float64)
            assert np.abs(got - c[key]).max() <= 1e-7 + 2e-7 * abs(c[key]), (i, c, mode, got.ravel()[:3])        # (fp32 rounding of an exact rational)
        elif f == "packed_position":
            assert close(means[i], c["expect"]), (i, c, means[i])
        elif f == "packed_scale":
            assert close(scales[i], c["expect"]), (i, c, scales[i])
        elif f == "packed_rotation":
            assert close(quats[i], c["expect"]), (i, c, quats[i])
        elif f == "packed_color":
            assert close(opac[i], c["expect_opacity"]) and (sh_dc is None or close(sh_dc[i], c["expect_dc"])), (i, c, opac[i])


def test_compressed_ply_known_answer_vectors(tmp_path):
    """Byte-level known answers for the PlayCanvas compressed layout (words written by hand, expected values in exact arithmetic:
    tests/golden/make_compressed_kat.py) through the NumPy decoder — from a FILE assembled here, so header parsing, element order and the
    `sh` element are in the loop — and through the payload reader the device path uses."""
    cases, chunks, packed, shb = _kat_payload()
    n = len(cases)
    path = tmp_path / "kat_compressed.ply"
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement chunk 1\n")
        for k in ply.CHUNK_PROPS + ply.CHUNK_COLOR_PROPS:
            f.write(f"property float {k}\n".encode())
        f.write(f"element vertex {n}\n".encode())
        for k in ply.PACKED_PROPS:
            f.write(f"property uint {k}\n".encode())
        f.write(f"element sh {n}\n".encode())
        for k in range(45):
            f.write(f"property uchar f_rest_{k}\n".encode())
        f.write(b"end_header\n")
        f.write(chunks.astype("<f4").tobytes()); f.write(packed.astype("<u4").tobytes()); f.write(shb.tobytes())
    m, s, q, o, sh, deg = ply.load_compressed_ply(str(path), sh_decode="bin_centre")
    assert deg == 3 and sh.shape == (n, 16, 3)
    _check_kat(cases, m, s, q, o, sh[:, 0, :], sh[:, 1:, :])
    for mode in ("linear255", "bin_centre_ends"):               # the other two readings of a coefficient byte
        _check_kat(cases, None, None, None, None, None, ply.load_compressed_ply(str(path), sh_decode=mode)[4][:, 1:, :], mode=mode)
    with pytest.raises(ValueError, match="sh_decode"):          # ... and no default: the caller has to say which
        ply.load_compressed_ply(str(path))
    c2, p2, b2, d2 = ply.read_compressed_payload(str(path))
    assert d2 == 3 and (c2 == chunks).all() and (p2 == packed).all() and (b2 == shb).all()


def test_compressed_encoder_round_trip_with_sh(tmp_path):
    """save_compressed_ply (degree 3: the `sh` element too) -> load_compressed_ply: every attribute within its quantisation step."""
    rng = np.random.default_rng(4)
    n = 1000
    means = rng.normal(size=(n, 3)).astype(np.float32) * 3
    scales = np.exp(rng.normal(-3.0, 0.7, (n, 3))).astype(np.float32)
    quats = rng.normal(size=(n, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
    opac = rng.uniform(0.02, 1.0, n).astype(np.float32)
    sh = (rng.normal(size=(n, 16, 3)) * 0.4).astype(np.float32)
    path = str(tmp_path / "c3.ply")
    ply.save_compressed_ply(path, means, scales, quats, opac, sh, 3)
    m, s, q, o, sh2, deg = ply.load_compressed_ply(path, sh_decode="bin_centre")
    assert deg == 3
    span = np.ptp(means, axis=0).max()
    assert np.abs(m - means).max() <= span / 1023 and np.abs(np.log(s / scales)).max() <= np.ptp(np.log(scales)) / 1023
    assert np.abs(o - opac).max() <= 0.5 / 255 + 1e-6 and np.abs(sh2[:, 1:] - np.clip(sh[:, 1:], -4, 4 - 8 / 256)).max() <= 4 / 256 + 1e-6
    assert np.abs(sh2[:, 0] - sh[:, 0]).max() <= 0.5 / 255 / ply.SH_C0 + 1e-5
    dot = np.abs(np.sum(q * quats, axis=1))
    assert dot.min() > 1 - 3e-6 * 1023                      # rotations agree to the 10-bit step


def test_sweep_batches_run_across_trajectories(tmp_path):
    """sweep.run cuts ONE work list per scene into GPU batches: three trajectories of 5, 7 and 4 waypoints with chunk = 8 are two batches
    of 8, every frame reaches on_frame (and the encoder) under its own trajectory / index / file name, in order, and a trajectory id that
    occurs twice is rendered once.  (A stand-in renderer that paints frame k of a batch with its camera's x position: no GPU.)"""
    import torch

    class Handle:
        def __init__(self, host):
            self.host = host

        def wait(self):
            return self.host

    class Ring:
        def __init__(self, shape):
            self.shape = shape

        def submit(self, buf, n):
            host = np.zeros(self.shape, np.uint8)
            host[:n, :, :, :3] = np.clip(buf[:n].numpy() * 255.0, 0, 255).astype(np.uint8)
            return Handle(host)

    class Fake:
        batches = []

        def host_frames(self, shape, depth=2):
            return Ring(shape)

        def render_batch(self, cams, scene, out=None):
            self.batches.append(len(cams))
            f = torch.zeros((len(cams), cams[0].height, cams[0].width, 3))
            for k, c in enumerate(cams):
                f[k] = float(np.linalg.inv(np.asarray(c.view, np.float64).reshape(4, 4))[0, 3]) / 100.0      # the camera's x, as a grey level
            return f

    def traj(tid, n, x0):
        return {"trajectory_id": tid, "instruction_index": 0,
                "points": [{"point": i, "position": [x0 + i, 0.0, 1.2], "rotation": [0.0, 0.0, 0.0, 1.0]} for i in range(n)]}
    trs = [traj("a", 5, 10), traj("b", 7, 30), traj("a", 5, 10), traj("c", 4, 60)]
    seen = []
    fake = Fake()
    n = sweep.run(fake, None, trs, "0042", str(tmp_path / "img"), resolution=(32, 24), chunk=8,
                  on_frame=lambda tid, i, rgb: seen.append((tid, i, int(rgb[0, 0, 0]))))
    assert fake.batches == [8, 8]
    assert [(t, i) for t, i, _ in seen] == [("a", i) for i in range(5)] + [("b", i) for i in range(7)] + [("c", i) for i in range(4)]
    x_of = {"a": 10, "b": 30, "c": 60}
    assert all(abs(v - round((x_of[t] + i) / 100.0 * 255.0)) <= 1 for t, i, v in seen)
    assert n == 21                                                    # (the metadata lists the repeated trajectory as the reference does)
    for t, cnt in (("a", 5), ("b", 7), ("c", 4)):
        assert sorted(os.listdir(tmp_path / "img" / "images" / f"trajectory_{t}")) == [f"0042_{t}_{i:03d}.jpg" for i in range(cnt)]
    # a second run finds every file and renders nothing
    fake.batches.clear()
    sweep.run(fake, None, trs, "0042", str(tmp_path / "img"), resolution=(32, 24), chunk=8)
    assert fake.batches == []


def test_sweep_writes_the_files_a_recorded_run_of_the_reference_wrote(tmp_path):
    """The reference's generator, run on a four-waypoint trajectory behind recording stand-ins (tests/golden/isaac_call_trace.json), wrote
    `<scene>/images/trajectory_<id>/<scene>_<traj>_<idx:03d>.jpg` + `<scene>/image_metadata.json`; sweep.run on the same sampled points
    must write the same file set, and every metadata field the two share must carry the same value (the reference adds the instruction
    text and its own processing notes, which come from files the sweep driver does not read)."""
    import torch
    trace = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "isaac_call_trace.json")))

    class Handle:
        def __init__(self, host):
            self.host = host

        def wait(self):
            return self.host

    class Fake:
        def host_frames(self, shape, depth=2):
            fake = self

            class Ring:
                def submit(self, buf, n):
                    host = np.zeros(shape, np.uint8)
                    host[:n, :, :, :3] = np.clip(buf[:n].numpy() * 255.0, 0, 255).astype(np.uint8)
                    return Handle(host)
            return Ring()

        def render_batch(self, cams, scene, out=None):
            return torch.zeros((len(cams), cams[0].height, cams[0].width, 3))
    actions = tmp_path / "action_groundtruth.json"
    json.dump({"groundtruth_data": [{"trajectory_id": "7", "instruction_index": 0, "sampled_points": trace["sampled_points"]}]}, open(actions, "w"))
    out = tmp_path / "out" / "0042"
    n = sweep.run(Fake(), None, sweep.load_trajectories(str(actions)), "0042", str(out), chunk=4)
    assert n == len(trace["sampled_points"])
    written = sorted(os.path.relpath(os.path.join(d, f), tmp_path / "out") for d, _, fs in os.walk(tmp_path / "out") for f in fs)
    assert written == trace["files_written"]
    ours, ref = json.load(open(out / "image_metadata.json")), trace["image_metadata"]
    for k in ("scene_id", "scene_name", "total_image_sequences", "frames_per_sequence", "image_resolution", "camera_settings"):
        assert ours[k] == ref[k], k
    so, sr = ours["sequences"][0], ref["sequences"][0]
    for k in ("scene_id", "trajectory_id", "instruction_index", "frame_filenames", "trajectory_sampled_points"):
        assert so[k] == sr[k], k
    assert so["sampling_info"]["sampled_points_count"] == sr["sampling_info"]["sampled_points_count"] == so["sampling_info"]["generated_images_count"]


def test_sh_byte_decode_modes_over_all_codes():
    """ply.decode_sh_bytes over the 256 codes: the three readings (include/sage_gs.h SGS_SH_DECODE_*) are increasing, agree to 1/64, invert
    the encoder's truncation bin (bin_centre re-encodes to the same byte), and the end codes of the other two are exactly -4 / +4."""
    v = np.arange(256, dtype=np.uint8)
    c, l, e = (ply.decode_sh_bytes(v, m) for m in ("bin_centre", "linear255", "bin_centre_ends"))
    assert c.dtype == l.dtype == e.dtype == np.float32
    for a in (c, l, e):
        assert (np.diff(a) > 0).all() and a.min() >= -4.0 and a.max() <= 4.0
    assert np.array_equal(c, (v.astype(np.float64) / 32.0 - 4.0 + 1.0 / 64.0).astype(np.float32))
    assert np.array_equal(np.clip(np.trunc((c.astype(np.float64) / 8.0 + 0.5) * 256.0), 0, 255).astype(np.uint8), v)      # encode(decode(v)) == v
    assert np.abs(l - c).max() <= 1.0 / 64.0 + 1e-7 and np.abs(e - c).max() == 1.0 / 64.0
    assert l[0] == -4.0 and l[255] == 4.0 and e[0] == -4.0 and e[255] == 4.0 and np.array_equal(e[1:255], c[1:255])
    with pytest.raises(ValueError):
        ply.decode_sh_bytes(v, "nearest")


def test_sweep_bounds_the_frames_waiting_for_the_encoder_and_always_shuts_its_pool_down(tmp_path, monkeypatch):
    """sweep.run's host side: at most 4 x encode_workers frames wait for (or are in) the JPEG encoder however fast the renderer produces them
    (every queued frame holds its own copy: unbounded, a scene of a few thousand waypoints held gigabytes), and the thread pool is shut down
    on every exit path — also when a callback raises half-way."""
    import concurrent.futures as cf
    import time
    import torch
    seen = {"max_pending": 0, "shutdowns": 0, "submitted": 0}

    class Pool(cf.ThreadPoolExecutor):
        def __init__(self, *a, **k):
            super().__init__(*a, **k); self._futs = []

        def submit(self, fn, *a, **k):
            self._futs = [f for f in self._futs if not f.done()]
            f = super().submit(fn, *a, **k)
            self._futs.append(f); seen["submitted"] += 1
            seen["max_pending"] = max(seen["max_pending"], len(self._futs))
            return f

        def shutdown(self, *a, **k):
            seen["shutdowns"] += 1
            return super().shutdown(*a, **k)
    monkeypatch.setattr(cf, "ThreadPoolExecutor", Pool)
    from PIL import Image
    real_save = Image.Image.save

    def slow_save(self, *a, **k):
        time.sleep(0.004)                       # an encoder far slower than the (fake) renderer
        return real_save(self, *a, **k)
    monkeypatch.setattr(Image.Image, "save", slow_save)

    class Handle:
        def __init__(self, host):
            self.host = host

        def wait(self):
            return self.host

    class Fake:
        def host_frames(self, shape, depth=2):
            class Ring:
                def submit(self, buf, n):
                    return Handle(np.zeros(shape, np.uint8))
            return Ring()

        def render_batch(self, cams, scene, out=None):
            return torch.zeros((len(cams), cams[0].height, cams[0].width, 3))
    traj = [{"trajectory_id": "t", "instruction_index": 0,
             "points": [{"point": i, "position": [0.1 * i, 0.0, 1.2], "rotation": [1.0, 0.0, 0.0, 0.0]} for i in range(96)]}]
    n = sweep.run(Fake(), None, traj, "0042", str(tmp_path / "a"), resolution=(32, 24), chunk=32, encode_workers=2)
    assert n == 96 and seen["submitted"] == 96 and seen["shutdowns"] == 1
    assert seen["max_pending"] <= 4 * 2 + 1, seen                       # (+1: the frame being submitted)
    assert len(os.listdir(tmp_path / "a" / "images" / "trajectory_t")) == 96

    def boom(tid, i, rgb):
        if i == 40:
            raise RuntimeError("callback failed")
    with pytest.raises(RuntimeError, match="callback failed"):
        sweep.run(Fake(), None, traj, "0042", str(tmp_path / "b"), resolution=(32, 24), chunk=32, encode_workers=2, on_frame=boom)
    assert seen["shutdowns"] == 2
