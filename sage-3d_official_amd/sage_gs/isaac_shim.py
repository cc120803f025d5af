"""The Isaac Sim side of the seam, as far as SAGE-3D's two render callers touch it.  Scope, precisely:
the CALL SET OF generate_images.py's frame generator (recorded from a run of the reference itself: tests/golden/isaac_call_trace.json,
replayed against this module by tests/test_next_rows.py) and the camera / world / stage calls of simple_env.py's render path listed
below run against the MI355X renderer without an edit.  NOT covered: the rest of simple_env.py's simulator surface — `Usd.PrimRange`,
`UsdGeom.Xformable`, the `UsdPhysics` collision APIs, `carb.settings`, `omni.replicator` / `omni.syntheticdata` / viewport modules
(simple_env.py:457, 585-670): `install()` registers `pxr.Usd` / `UsdPhysics` / `Sdf` as EMPTY namespaces so that import lines
resolve, and a call into them raises AttributeError.  SimpleVLNEnv as a whole therefore does not construct against this shim; its
get_rgb()-shaped callers are served through adapter.GsCamera directly.

    reference call                                              (file:line)                          here

    reference call                                              (file:line)                          here
    ----------------------------------------------------------  -----------------------------------  ---------------------------------
    SimulationApp({"headless": True})                           generate_images.py:27, simple_env:163  SimulationApp: holds the config
    omni.usd.get_context().close_stage() / .get_stage()         generate_images.py:321,328; se:219-221 UsdContext
    open_stage(usd_path=...) -> bool                            generate_images.py:324; se:220         open_stage: .usda -> asset -> scene
    stage.GetPrimAtPath(path) / UsdLux.DomeLight.Define(...)    generate_images.py:331-334             Stage / no-op lights
    World(); world.reset(); world.step(render=True); .clear()   generate_images.py:338-341,426; se:225 World: nothing to step
    Camera(prim_path=, frequency=, resolution=); .initialize()  generate_images.py:345-346; se:840-872 Camera -> adapter.GsCamera
    UsdGeom.Camera(prim).GetFocalLengthAttr().Set(8.0)          generate_images.py:348-350; se:905     -> GsCamera.set_focal_length
    cam.set_world_pose(position=, orientation=); cam.get_rgba() generate_images.py:419-428; se:1284,1380  GsCamera

`install()` registers these under the module names the reference imports (`omni.usd`, `omni.isaac.core`,
`omni.isaac.core.utils.stage`, `omni.isaac.sensor`, `omni.isaac.kit`, `isaacsim.simulation_app`, `isaacsim.sensors.camera`,
`pxr`), so its import lines resolve; code written against this module directly imports the names from here.

`open_stage(path)` resolves a SAGE-3D scene stage: `{scene_id}.usda` -> the `@...usdz[gauss.usda]@` reference of /World/gauss
(adapter.parse_scene_usda; sage3d_usda_builder.py:93-149) -> the Gaussians beside that USDZ — `<stem>.ply`, `<dir>/3dgs.ply`, or the
PlayCanvas-compressed `<stem>_compressed.ply` / `<dir>/3dgs_compressed.ply` the USDZ was converted from (README.md:210-253) — with the
prim's model->world transform (template.usda:115-124); a `.ply` path is opened as such.  The scene is uploaded once per stage, as
the reference loads a stage once per scene.  Lights, physics and collision payloads are accepted and ignored: the renderer produces
linear radiance from the Gaussians alone.  Failures raise (no silent `False` for a missing asset: `open_stage`
returns False only where the reference's would — the stage file does not exist).
"""
from __future__ import annotations

import os
import sys
import types
from typing import Optional, Tuple

import numpy as np

from . import adapter

_state = {"renderer": None, "loader": None, "stage": None, "app": None, "scene_root": None}


def configure(renderer=None, loader=None, scene_root=None):
    """renderer: the sage_gs.Renderer frames are drawn with (default: one on cuda:LOCAL_RANK / cuda:0, created at the first
    open_stage).  loader(path, compressed) -> (means, scales, quats, opacities, sh, degree) (default: sage_gs.ply).  scene_root: an
    extra directory searched for `<scene_id>.ply` / `<scene_id>_compressed.ply` when nothing lies beside the referenced USDZ."""
    if renderer is not None:
        _state["renderer"] = renderer
    if loader is not None:
        _state["loader"] = loader
    if scene_root is not None:
        _state["scene_root"] = scene_root


def _renderer():
    if _state["renderer"] is None:
        from .renderer import Renderer
        _state["renderer"] = Renderer(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    return _state["renderer"]


def _load(path, compressed):
    if _state["loader"] is not None:
        return _state["loader"](path, compressed)
    from . import ply
    return (ply.load_compressed_ply if compressed else ply.load_ply)(path)


class SimulationApp:
    """omni.isaac.kit.SimulationApp / isaacsim.simulation_app.SimulationApp: there is no application to launch."""

    def __init__(self, config: Optional[dict] = None):
        self.config = dict(config or {})
        _state["app"] = self

    def update(self):
        return None

    def is_running(self):
        return True

    def close(self):
        ctx = get_context()
        ctx.close_stage()
        _state["app"] = None


class Prim:
    def __init__(self, stage, path, kind="Xform"):
        self._stage, self._path, self.kind = stage, path, kind
        self.attrs = {}

    def IsValid(self):
        return True

    def GetPath(self):
        return self._path

    def __bool__(self):
        return True


class _Attr:
    def __init__(self, setter=None, value=None):
        self._set, self._v = setter, value

    def Set(self, v):
        self._v = v
        if self._set:
            self._set(v)
        return True

    def Get(self):
        return self._v


class Stage:
    """What `omni.usd.get_context().get_stage()` hands the callers: the prims they ask about, and the scene the stage holds."""

    def __init__(self, path, parsed, scene, model_to_world, asset):
        self.path, self.parsed, self.scene, self.model_to_world, self.asset = path, parsed, scene, model_to_world, asset
        self._prims = {"/World": Prim(self, "/World"), "/World/gauss": Prim(self, "/World/gauss")}
        self.cameras = {}

    def GetPrimAtPath(self, path):
        path = str(path)
        return self._prims.get(path) or _NoPrim(path)

    def DefinePrim(self, path, kind="Xform"):
        p = self._prims[str(path)] = Prim(self, str(path), kind)
        return p

    def RemovePrim(self, path):
        self._prims.pop(str(path), None)
        self.cameras.pop(str(path), None)
        return True


class _NoPrim:
    def __init__(self, path):
        self._path = path

    def IsValid(self):
        return False

    def __bool__(self):
        return False


class UsdContext:
    def get_stage(self):
        return _state["stage"]

    def close_stage(self):
        st = _state["stage"]
        if st is not None and st.scene is not None and hasattr(st.scene, "free"):
            st.scene.free()
        _state["stage"] = None
        return True

    def open_stage(self, path):
        return open_stage(path)


_context = UsdContext()


def get_context():
    return _context


def find_gaussians(usda_path: str, parsed: dict) -> Tuple[Optional[str], bool]:
    """(path, compressed?) of the Gaussians a scene stage refers to, or (None, False).  Searched: beside the referenced USDZ
    (`<stem>.ply`, `<stem>_compressed.ply`, `<dir>/3dgs.ply`, `<dir>/3dgs_compressed.ply`), then scene_root/<scene_id>[…]."""
    cands = []
    base = os.path.dirname(os.path.abspath(usda_path))
    usdz = parsed.get("usdz")
    if usdz:
        z = usdz if os.path.isabs(usdz) else os.path.normpath(os.path.join(base, usdz))
        stem, zdir = os.path.splitext(z)[0], os.path.dirname(z)
        cands += [(stem + ".ply", False), (stem + "_compressed.ply", True), (os.path.join(zdir, "3dgs.ply"), False),
                  (os.path.join(zdir, "3dgs_compressed.ply"), True)]
    sid = os.path.splitext(os.path.basename(usda_path))[0]
    for root in (_state["scene_root"], base):
        if root:
            cands += [(os.path.join(root, sid + ".ply"), False), (os.path.join(root, sid + "_compressed.ply"), True),
                      (os.path.join(root, sid, "3dgs.ply"), False), (os.path.join(root, sid, "3dgs_compressed.ply"), True)]
    for path, comp in cands:
        if os.path.isfile(path):
            return path, comp
    return None, False


def open_stage(usd_path: str) -> bool:
    """omni.isaac.core.utils.stage.open_stage: False when the stage file does not exist (as the reference's), otherwise the scene
    is resolved, loaded and uploaded — and anything wrong with it raises."""
    usd_path = str(usd_path)
    if not os.path.isfile(usd_path):
        return False
    get_context().close_stage()
    if usd_path.lower().endswith(".ply"):
        parsed, asset, comp = {}, usd_path, usd_path.lower().endswith("_compressed.ply")
        m2w = np.asarray(__import__("sage_gs.scenes", fromlist=["MODEL_TO_WORLD"]).MODEL_TO_WORLD, np.float64)
    else:
        parsed = adapter.parse_scene_usda(open(usd_path, "r", encoding="utf-8", errors="replace").read())
        asset, comp = find_gaussians(usd_path, parsed)
        if asset is None:
            raise FileNotFoundError(f"{usd_path}: /World/gauss references {parsed.get('usdz')!r}, and no Gaussian .ply was found beside it "
                                    "(looked for <stem>.ply, <stem>_compressed.ply, 3dgs.ply, 3dgs_compressed.ply; isaac_shim.configure(scene_root=...))")
        m2w = adapter.asset_model_to_world(parsed)
    arrays = _load(asset, comp)
    r = _renderer()
    from . import ply
    scene = r.upload(ply.to_gaussians(arrays, r.device, m2w))
    _state["stage"] = Stage(usd_path, parsed, scene, m2w, asset)
    return True


class World:
    """omni.isaac.core.World: nothing is simulated here — a frame is rendered when the camera is read."""
    _instance = None

    def __init__(self, *args, **kwargs):
        World._instance = self
        self.steps = 0
        self.renders = 0

    @classmethod
    def instance(cls):
        return cls._instance

    def reset(self, *args, **kwargs):
        return None

    def step(self, render: bool = True, *args, **kwargs):
        self.steps += 1
        self.renders += 1 if render else 0
        return None

    def render(self):
        self.renders += 1

    def play(self):
        return None

    def pause(self):
        return None

    def stop(self):
        return None

    def clear(self):
        return None

    def is_playing(self):
        return True

    def get_physics_dt(self):
        return 1.0 / 60.0

    @classmethod
    def clear_instance(cls):
        cls._instance = None


def Camera(prim_path: str = "/World/Camera", frequency: int = 30, resolution: Tuple[int, int] = (640, 480), **kwargs):
    """omni.isaac.sensor.Camera / isaacsim.sensors.camera.Camera(prim_path, frequency, resolution): a GsCamera on the open stage's
    scene (simple_env.py:840-844; generate_images.py:345).  Extra keyword arguments the simulator takes (position, orientation, dt, name)
    are honoured where they mean something here (position / orientation) and otherwise ignored."""
    st = _state["stage"]
    if st is None:
        raise RuntimeError("Camera(): no stage is open (call open_stage first)")
    cam = adapter.GsCamera(_renderer(), st.scene, prim_path=str(prim_path), frequency=int(frequency), resolution=tuple(resolution))
    if "position" in kwargs or "orientation" in kwargs:
        cam.set_world_pose(kwargs.get("position"), kwargs.get("orientation"))
    st.DefinePrim(str(prim_path), "Camera")
    st.cameras[str(prim_path)] = cam
    return cam


class _UsdGeomCamera:
    """pxr.UsdGeom.Camera(prim): the attributes the callers set on the camera prim (focal length, clipping range, apertures)."""

    def __init__(self, prim):
        st = _state["stage"]
        self._cam = st.cameras.get(prim.GetPath()) if (st is not None and prim) else None
        self._aperture = 20.955
        self._focal = 8.0 if self._cam is None else self._cam._focal_over_aperture * 20.955

    def _apply(self):
        if self._cam is not None:
            self._cam.set_focal_length(self._focal, self._aperture)

    def GetFocalLengthAttr(self):
        def s(v):
            self._focal = float(v); self._apply()
        return _Attr(s, self._focal)

    def GetHorizontalApertureAttr(self):
        def s(v):
            self._aperture = float(v); self._apply()
        return _Attr(s, self._aperture)

    def GetVerticalApertureAttr(self):
        return _Attr(None, self._aperture * 0.75)

    def GetClippingRangeAttr(self):
        return _Attr(None, (0.1, 50.0))                         # simple_env.py:899 sets (0.1, 50): the renderer's near plane is its own


class _Light:
    def CreateIntensityAttr(self, v=None):
        return _Attr(None, v)

    def CreateColorAttr(self, v=None):
        return _Attr(None, v)

    def CreateTextureFileAttr(self, v=None):
        return _Attr(None, v)


class _DomeLight:
    @staticmethod
    def Define(stage, path):
        stage.DefinePrim(path, "DomeLight")
        return _Light()


UsdGeom = types.SimpleNamespace(Camera=_UsdGeomCamera)
UsdLux = types.SimpleNamespace(DomeLight=_DomeLight, DistantLight=_DomeLight, SphereLight=_DomeLight)
Gf = types.SimpleNamespace(Vec3f=lambda *a: tuple(float(v) for v in a), Vec3d=lambda *a: tuple(float(v) for v in a),
                           Vec2f=lambda *a: tuple(float(v) for v in a))


def install(force: bool = False):
    """Register the shim under the module names the reference imports, unless the real ones are importable (force=True overrides):
    after this, `from omni.isaac.core import World`, `from omni.isaac.core.utils.stage import open_stage`,
    `from omni.isaac.sensor import Camera`, `import omni.usd`, `from pxr import Gf, UsdGeom, UsdLux` and either SimulationApp import
    resolve to this module's objects (generate_images.py:24-33; simple_env.py:28-31,163-179)."""
    if not force:
        try:
            import omni.usd  # noqa: F401
            return False                      # a real Isaac Sim is present: leave it alone
        except Exception:
            pass

    made = {}

    def mod(name, **attrs):
        m = made.get(name)
        if m is None:
            m = None if force else sys.modules.get(name)
            if m is None:
                m = types.ModuleType(name)
                m.__path__ = []               # a package: sub-modules may hang below it
                sys.modules[name] = m
            made[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(mod(parent), leaf, m)
        return m

    mod("omni.usd", get_context=get_context)
    mod("omni.isaac.kit", SimulationApp=SimulationApp)
    mod("omni.isaac.core", World=World)
    mod("omni.isaac.core.utils.stage", open_stage=open_stage)
    mod("omni.isaac.core.utils.extensions", enable_extension=lambda *a, **k: True)
    mod("omni.isaac.sensor", Camera=Camera)
    mod("omni.kit.commands", execute=lambda *a, **k: (True, None))
    mod("isaacsim.simulation_app", SimulationApp=SimulationApp)
    mod("isaacsim.sensors.camera", Camera=Camera)
    mod("isaacsim.core.utils.extensions", enable_extension=lambda *a, **k: True)
    mod("pxr", Gf=Gf, UsdGeom=UsdGeom, UsdLux=UsdLux,
        UsdPhysics=types.SimpleNamespace(), Usd=types.SimpleNamespace(), Sdf=types.SimpleNamespace(Path=str))
    return True
