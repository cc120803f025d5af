#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run in THIS container only (it imports the reference).

  pose_golden.json   — inputs/outputs of the reference's own importable pose functions
                       (Code/data_pipeline/trajectory_generation/trajectory_2d_to_3d.py:
                        quaternion_from_yaw :79, yaw_from_quaternion :66, transform_trajectory_points :124;
                        Code/data_pipeline/training_data_construction/generate_actions.py:
                        BatchActionGenerator.yaw_from_quaternion :117).  The reference holds no tests, so these
                       are the only results of the reference itself that can be pinned for this path.
  pose_env_golden.json — the benchmark environment's pose branch: SimpleVLNEnv.set_start_pose (simple_env.py:1149-1195) and
                       _update_camera_position (:1196-1317) run AS THEY ARE on an instance made without Isaac Sim (stub
                       modules stand in for `imageio` / `isaacsim` at import time only; a recording object stands in for
                       the camera) — captured: the yaw the environment derives and the (position, orientation) it hands to
                       cam.set_world_pose, at the start pose and after yaw changes.
  usda_golden.json   — the shape of a scene stage built by the reference's own builder from Data/template.usda, as
                       (type, name, value) triples of the two prims the render path reads (see usda_fixture)
  config1_golden.npz — a small BASELINE config-1 frame from the fp64 NumPy oracle (image, tile offsets,
                       queue order, per-splat rects): pins the oracle against regressions and gives the GPU
                       tests a fixture that does not depend on running the oracle.
  isaac_call_trace.json — the reference's frame generator itself (generate_images.SequentialFastImageGenerator.process_single_file,
                       generate_images.py:292-560) RUN here on a four-waypoint trajectory behind RECORDING stand-ins for the Isaac Sim
                       modules it imports: every call it makes on SimulationApp / omni.usd / open_stage / World / Camera / pxr, in order,
                       with its arguments — plus the files it wrote.  tests/test_next_rows.py replays that sequence against
                       sage_gs.isaac_shim: the protocol the shim must serve is DATA recorded from the reference, not a transcription.
Only DATA is written; no reference source text is copied.
"""
import copy
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Code"


def pose_fixture():
    sys.path.insert(0, os.path.join(REF, "data_pipeline", "trajectory_generation"))
    sys.path.insert(0, os.path.join(REF, "data_pipeline", "training_data_construction"))
    import trajectory_2d_to_3d as t23
    rng = np.random.default_rng(42)
    yaws = [0.0, 0.3, -0.3, 1.0, -2.5, 3.0, -3.1, math.pi / 2, -math.pi / 2] + list(rng.uniform(-math.pi, math.pi, 16))
    cases = []
    for yaw in yaws:
        q = t23.quaternion_from_yaw(yaw)                      # the pre-transform quaternion (0,0,qz,qw)
        pts = [{"position": [float(rng.uniform(0, 5)), float(rng.uniform(0, 5)), 0.0], "rotation": list(q)},
               {"position": [1.0, 2.0, 0.0], "rotation": list(q)}]      # the LAST point's rotation is reset by the reference
        before = copy.deepcopy(pts)
        t23.transform_trajectory_points(pts, 0.0, 5.0, 0.0, 5.0)
        cases.append({"yaw": float(yaw), "quaternion_from_yaw": [float(v) for v in q],
                      "yaw_from_quaternion": float(t23.yaw_from_quaternion(*q)),
                      "points_before": before, "points_after": pts})
    # the action generator's decoder on the transformed rotations
    try:
        import generate_actions as ga
        dec = ga.BatchActionGenerator.yaw_from_quaternion
        for c in cases:
            c["action_generator_yaw"] = float(dec(None, c["points_after"][0]["rotation"]))
    except Exception as e:                                   # pragma: no cover
        print("generate_actions not importable:", e)
    json.dump({"source": "Galery23/SAGE-3D_Official @ /root/reference (functions named in make_golden.py)", "cases": cases},
              open(os.path.join(HERE, "pose_golden.json"), "w"), indent=1)
    print("pose_golden.json:", len(cases), "cases")


def pose_env_fixture():
    """pose_env_golden.json: inputs -> what SimpleVLNEnv passes to cam.set_world_pose.  Nothing of the simulator runs: the
    two methods are pure arithmetic on self._pos / self._yaw / self._original_quaternion / self._initial_yaw."""
    import contextlib
    import io
    import types
    stubs = {}
    for name in ("imageio", "isaacsim", "isaacsim.simulation_app"):
        if name not in sys.modules:
            stubs[name] = sys.modules[name] = types.ModuleType(name)
    sys.modules["isaacsim.simulation_app"].SimulationApp = object
    sys.path.insert(0, os.path.join(REF, "benchmark", "environment_evaluation"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import simple_env
    finally:
        for name in stubs:
            sys.modules.pop(name, None)

    class RecCam:
        def __init__(self): self.calls = []
        def set_world_pose(self, position=None, orientation=None):
            self.calls.append(([float(v) for v in position], [float(v) for v in orientation],
                               str(np.asarray(position).dtype), str(np.asarray(orientation).dtype)))

    class NoWorld:
        def step(self, render=True): pass

    def fresh():
        env = object.__new__(simple_env.SimpleVLNEnv)
        env._log_function = None
        env.cam = RecCam()
        env.world = NoWorld()
        return env

    rng = np.random.default_rng(7)
    yaws = [0.0, 0.3, -0.3, 1.0, -2.5, 3.0, -3.1, math.pi / 2, -math.pi / 2] + list(rng.uniform(-math.pi, math.pi, 7))
    sys.path.insert(0, os.path.join(REF, "data_pipeline", "trajectory_generation"))
    import trajectory_2d_to_3d as t23
    cases = []
    for yaw in yaws:
        # a trajectory rotation exactly as the reference's own pipeline writes it (transform_trajectory_points)
        pts = [{"position": [0.0, 0.0, 0.0], "rotation": list(t23.quaternion_from_yaw(float(yaw)))},
               {"position": [1.0, 1.0, 0.0], "rotation": list(t23.quaternion_from_yaw(float(yaw)))}]
        t23.transform_trajectory_points(pts, 0.0, 5.0, 0.0, 5.0)
        rot = [float(v) for v in pts[0]["rotation"]]
        pos = [float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), float(rng.uniform(0, 2))]
        env = fresh()
        with contextlib.redirect_stdout(io.StringIO()):
            env.set_start_pose(list(pos), list(rot))
        start = env.cam.calls[-1]
        steps = []
        for d in (0.004, 0.02, -0.35, 1.3, float(rng.uniform(-3, 3))):     # yaw changes: below / above the 0.01 rad switch
            env._yaw = env._initial_yaw + d                                 # (what the step loop does, simple_env.py:2054)
            env._pos = np.array([pos[0] + 0.25 * d, pos[1] - 0.5 * d, pos[2]], dtype=np.float32)
            with contextlib.redirect_stdout(io.StringIO()):
                env._update_camera_position()
            c = env.cam.calls[-1]
            steps.append({"yaw": float(env._yaw), "agent_position": [float(v) for v in env._pos],
                          "camera_position": c[0], "orientation": c[1]})
        cases.append({"trajectory_yaw": float(yaw), "position": pos, "rotation_xyzw": rot,
                      "start_yaw": float(env._initial_yaw), "start_camera_position": start[0], "start_orientation": start[1],
                      "dtypes": [start[2], start[3]], "steps": steps})
    # the fallback branch (no trajectory quaternion known): orientation from the yaw alone
    fb = []
    for yaw in (0.0, 0.7, -2.0):
        env = fresh(); env._pos = np.array([1.0, 2.0, 0.3], dtype=np.float32); env._yaw = yaw
        with contextlib.redirect_stdout(io.StringIO()):
            env._update_camera_position()
        fb.append({"yaw": yaw, "camera_position": env.cam.calls[-1][0], "orientation": env.cam.calls[-1][1]})
    json.dump({"source": "SimpleVLNEnv.set_start_pose / _update_camera_position of Galery23/SAGE-3D_Official run by make_golden.py "
                         "(no simulator: stub imports, recording camera)", "cases": cases, "fallback": fb},
              open(os.path.join(HERE, "pose_env_golden.json"), "w"), indent=1)
    print("pose_env_golden.json:", len(cases), "cases x", len(cases[0]["steps"]), "steps,", len(fb), "fallback")


def usda_fixture():
    """usda_golden.json — the SHAPE of a scene stage as the reference builds it, as data.  The reference's own
    `sage3d_usda_builder.build_usda_content` (Code/benchmark/scene_data/sage3d_usda_builder.py:93-149) is run here on
    Data/template.usda for one scene id; from its output only the two prims the render path reads are tokenised —
    /World/gauss (template.usda:115-124) and /World/scene_collision (:156-165) — into (type, name, value) triples and
    composition arcs, plus the stage's upAxis / metersPerUnit.  No template text is stored: the test re-serialises the
    triples itself and checks `sage_gs.adapter.parse_scene_usda` on the result."""
    import re
    sys.path.insert(0, os.path.join(REF, "benchmark", "scene_data"))
    import sage3d_usda_builder as b
    template = open("/root/reference/Data/template.usda", "r", encoding="utf-8").read()
    scene_id = "0042"
    usdz_t = "@/data/InteriorGS_usdz/{scene_id}.usdz[gauss.usda]@"
    coll_t = "@/data/InteriorGS_Collision/Collision/{scene_id}/{scene_id}_collision.usd@"
    text = b.build_usda_content(template, scene_id, "839920", "@usdz_root[gauss.usda]@", usdz_t, "@collision_root@", coll_t)

    def prim(header_re):
        m = re.search(header_re, text, re.S)
        meta = re.search(r"\((.*?)\)", text[m.start():], re.S).group(1)
        arcs = [[a.strip(), v.strip()] for a, v in re.findall(r"(prepend\s+\w+)\s*=\s*(@[^@]*@)", meta)]
        i = text.find("{", m.end())
        j = text.find("}", i)
        attrs = []
        for line in text[i + 1:j].splitlines():
            line = line.strip()
            mm = re.match(r"((?:uniform\s+)?[\w\[\]]+)\s+([\w:]+)\s*=\s*(.+)$", line)
            if mm:
                attrs.append([mm.group(1), mm.group(2), mm.group(3)])
        return {"arcs": arcs, "attrs": attrs}

    out = {"source": "sage3d_usda_builder.build_usda_content on Data/template.usda (reference), tokenised by make_golden.py",
           "scene_id": scene_id, "usdz_path_template": usdz_t, "collision_path_template": coll_t,
           "stage": {"upAxis": re.search(r'upAxis\s*=\s*"(\w)"', text).group(1),
                     "metersPerUnit": float(re.search(r"metersPerUnit\s*=\s*([\d.]+)", text).group(1))},
           "gauss": dict(prim(r'over\s+"gauss"'), specifier='over "gauss"'),
           "scene_collision": dict(prim(r'def\s+"scene_collision"'), specifier='def "scene_collision"'),
           "authoring_layer": re.search(r'authoring_layer\s*=\s*"([^"]+)"', text).group(1)}
    json.dump(out, open(os.path.join(HERE, "usda_golden.json"), "w"), indent=1)
    print("usda_golden.json:", out["gauss"]["arcs"], len(out["gauss"]["attrs"]), "attrs;", out["scene_collision"]["arcs"])


def isaac_trace_fixture():
    """isaac_call_trace.json: what the reference's image generator asks of Isaac Sim, recorded by running it."""
    import contextlib
    import io
    import tempfile
    import types
    events = []

    def js(v):
        if isinstance(v, np.ndarray):
            return {"ndarray": v.astype(np.float64).tolist(), "dtype": str(v.dtype)}
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        if isinstance(v, (list, tuple)):
            return [js(x) for x in v]
        if isinstance(v, dict):
            return {str(k): js(x) for k, x in v.items()}
        if isinstance(v, Rec):
            return {"ref": v._name}
        if isinstance(v, (str, int, float, bool)) or v is None:
            return v
        return {"repr": type(v).__name__}

    class Rec:
        """an object that records every call made on it (and on what those calls return)"""
        _results = {}

        def __init__(self, name):
            object.__setattr__(self, "_name", name)

        def __getattr__(self, attr):
            return Rec(f"{self._name}.{attr}")

        def __call__(self, *a, **kw):
            events.append({"call": self._name, "args": js(list(a)), "kwargs": js(kw)})
            res = Rec._results.get(self._name, None)
            if callable(res):
                return res(*a, **kw)
            if self._name in Rec._results:
                return res
            return Rec(self._name + "()")

        def __bool__(self):
            return True

    frame_no = [0]

    def fake_rgba():
        frame_no[0] += 1
        img = np.zeros((768, 1024, 4), np.uint8); img[..., 3] = 255; img[..., 0] = frame_no[0]
        return img
    Rec._results = {
        "open_stage": lambda **kw: os.path.exists(kw.get("usd_path", "")),
        "Camera().get_rgba": fake_rgba,
        "omni.usd.get_context().get_stage().GetPrimAtPath": lambda path: Rec("prim") if path != "/World/EnvLight" else None,
    }
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); mods[name] = m
        return m
    mod("omni"); mod("omni.isaac"); mod("omni.isaac.core.utils")
    mod("omni.isaac.kit", SimulationApp=Rec("SimulationApp"))
    mod("omni.usd", get_context=Rec("omni.usd.get_context"))
    mod("omni.isaac.core", World=Rec("World"))
    mod("omni.isaac.core.utils.stage", open_stage=Rec("open_stage"))
    mod("omni.isaac.sensor", Camera=Rec("Camera"))
    mod("pxr", Gf=Rec("Gf"), UsdGeom=Rec("UsdGeom"), UsdLux=Rec("UsdLux"))
    mods["omni"].usd = mods["omni.usd"]
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    sys.path.insert(0, os.path.join(REF, "data_pipeline", "training_data_construction"))
    points = [{"point_id": i, "position": [1.0 + 0.1 * i, 2.0, 0.3], "rotation": [math.cos(0.2 * i), 0.0, 0.0, math.sin(0.2 * i)]} for i in range(4)]
    try:
        with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
            import generate_images as gi
            os.makedirs(os.path.join(td, "traj", "0042")); os.makedirs(os.path.join(td, "usda")); os.makedirs(os.path.join(td, "actions", "0042"))
            open(os.path.join(td, "usda", "0042.usda"), "w").write("#usda 1.0\n")
            json.dump({"scenes": [{"scene_id": "0042", "samples": [{"trajectory_id": "7", "points": [], "instructions": ["go to the sofa"]}]}]},
                      open(os.path.join(td, "traj", "0042", "train_trajectories_0042.json"), "w"))
            json.dump({"groundtruth_data": [{"trajectory_id": "7", "instruction_index": 0, "sampled_points": points}]},
                      open(os.path.join(td, "actions", "0042", "action_groundtruth.json"), "w"))
            gen = gi.SequentialFastImageGenerator(os.path.join(td, "traj"), os.path.join(td, "usda"), os.path.join(td, "out"),
                                                  action_root=os.path.join(td, "actions"), force=True)
            files = gen.find_all_trajectory_files()
            n_import = len(events)                 # (SimulationApp({...}) happens at import time)
            ok = gen.process_single_file(files[0])
            written = sorted(os.path.relpath(os.path.join(d, f), os.path.join(td, "out")) for d, _, fs in os.walk(os.path.join(td, "out")) for f in fs)
            meta = json.load(open(os.path.join(td, "out", "0042", "image_metadata.json")))
            # paths inside the trace are the temp dir's: keep them relative
            for e in events:
                for k, v in list(e["kwargs"].items()):
                    if isinstance(v, str) and v.startswith(td):
                        e["kwargs"][k] = os.path.relpath(v, td)
            consts = {"CAMERA_RESOLUTION": list(gi.CAMERA_RESOLUTION), "CAMERA_FOCAL_LENGTH": gi.CAMERA_FOCAL_LENGTH, "CAMERA_HEIGHT": gi.CAMERA_HEIGHT,
                      "WORLD_STEP_COUNT": gi.WORLD_STEP_COUNT, "RENDER_STEP_COUNT": gi.RENDER_STEP_COUNT}
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("generate_images", None)
    assert ok, "the reference's process_single_file reported failure"

    def strip(v):                  # the metadata as data: drop absolute paths and timestamps
        if isinstance(v, dict):
            return {k: strip(x) for k, x in v.items() if k != "source_files" and not any(t in k.lower() for t in ("time", "path", "dir"))}
        if isinstance(v, list):
            return [strip(x) for x in v]
        return v
    json.dump({"source": "generate_images.SequentialFastImageGenerator.process_single_file (generate_images.py:292-560) run behind recording stand-ins "
                         "for the Isaac Sim modules; tests/golden/make_golden.py isaac_trace_fixture",
               "constants": consts, "sampled_points": points, "events_at_import": n_import, "events": events,
               "files_written": written, "image_metadata": strip(meta)},
              open(os.path.join(HERE, "isaac_call_trace.json"), "w"), indent=1)
    print("isaac_call_trace.json:", len(events), "calls,", len(written), "files written")


def config1_fixture():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_np as onp
    scene, _ = onp.config1_scene(n=2500, seed=0)
    cam = onp.Camera(128, 128, 64.0, 64.0, 64.0, 64.0, np.eye(4, dtype=np.float32))
    img, aux = onp.render(*scene, cam)
    pre = aux["pre"]
    # two-sided answers for the threshold-sensitive pixels (margin < 1e-4): every colour an admissible evaluation gives
    ys, xs = np.nonzero(aux["margin"] < 1.0e-4)
    var, ptr = [], [0]
    for y, x in zip(ys, xs):
        v = onp.pixel_variants(pre, aux["offsets"], aux["ids"], cam, onp.Config(), int(x), int(y), 1.0e-4)
        var.append(v); ptr.append(ptr[-1] + len(v))
    np.savez_compressed(os.path.join(HERE, "config1_golden.npz"),
                        n=2500, seed=0, width=128, height=128, f=64.0,
                        image=img.astype(np.float32), margin=aux["margin"].astype(np.float32),
                        flag_yx=np.stack([ys, xs], 1).astype(np.int32), flag_ptr=np.asarray(ptr, np.int32),
                        flag_rgb=(np.concatenate(var) if var else np.zeros((0, 3))).astype(np.float32),
                        offsets=aux["offsets"].astype(np.int32), ids=aux["ids"].astype(np.int32),
                        rect=pre["rect"].astype(np.int16), tiles=pre["tiles"].astype(np.int32),
                        depth_bits=pre["depth"].view(np.uint32), n_contrib=aux["n_contrib"].astype(np.int32),
                        n_visible=aux["n_visible"], D=aux["D"], D_f=aux["D_f"])
    print("config1_golden.npz: D =", aux["D"], "D_f =", aux["D_f"], "threshold-sensitive pixels:", len(ys), "variants:", ptr[-1])


if __name__ == "__main__":
    if "--frames-only" not in sys.argv:          # (the pose / usda fixtures import the reference; the frame fixture does not)
        pose_fixture()
        pose_env_fixture()
        usda_fixture()
        isaac_trace_fixture()
    if "--reference-only" not in sys.argv:
        config1_fixture()
