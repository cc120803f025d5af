"""`GsCamera` — the Isaac Sim Camera protocol, as far as SAGE-3D's callers use it.

`simple_env.py:840-907,1284,1380` and `generate_images.py:344-350,417-432` construct
`Camera(prim_path, frequency, resolution)`, call `.initialize()`, `.set_world_pose(position, orientation)`
/ `.get_world_pose()`, and read `.get_rgba()` (uint8 [H,W,4]) after stepping the world.  This object
offers the same calls on top of the MI355X renderer, so those loops can be pointed at it unchanged:
there is no world to step — `get_rgba()` renders on demand.  Unlike the reference's callers' expectations
of Isaac Sim, failures raise (no silent `None` / black frames).
"""
from __future__ import annotations

import re
from typing import Optional, Sequence, Tuple

import numpy as np

from . import camera as cam_conv
from .renderer import Camera, RenderConfig, Renderer


class GsCamera:
    def __init__(self, renderer: Renderer, scene, prim_path: str = "/World/Camera", frequency: int = 30,
                 resolution: Tuple[int, int] = (640, 480), config: Optional[RenderConfig] = None):
        self._r, self._scene, self.prim_path, self.frequency = renderer, scene, prim_path, frequency
        self._w, self._h = int(resolution[0]), int(resolution[1])         # Isaac resolution = (width, height)
        self._pos = np.zeros(3, np.float32)
        self._orient = np.array([1.0, 0.0, 0.0, 0.0], np.float32)         # (w, x, y, z)
        self._focal_over_aperture = cam_conv.REF_FOCAL_OVER_APERTURE      # focalLength 8.0 (simple_env.py:905)
        self._config = config
        self._initialized = False

    # -- the protocol ---------------------------------------------------------------------------------
    def initialize(self):
        self._initialized = True

    def set_world_pose(self, position: Sequence[float] = None, orientation: Sequence[float] = None):
        if position is not None:
            self._pos = np.asarray(position, np.float32).reshape(3).copy()
        if orientation is not None:
            self._orient = np.asarray(orientation, np.float32).reshape(4).copy()

    def get_world_pose(self):
        return self._pos.copy(), self._orient.copy()

    def set_focal_length(self, focal_length_mm: float, horizontal_aperture_mm: float = 20.955):
        self._focal_over_aperture = float(focal_length_mm) / float(horizontal_aperture_mm)

    def get_resolution(self):
        return self._w, self._h

    def _camera(self) -> Camera:
        f = self._w * self._focal_over_aperture
        return Camera(self._w, self._h, f, f, self._w / 2.0, self._h / 2.0,
                      cam_conv.view_from_isaac_pose(self._pos, self._orient))

    def get_rgb_tensor(self):
        """float32 [H,W,3] on the GPU (linear RGB) — for consumers that stay on the device."""
        return self._r.render(self._camera(), self._scene, config=self._config)

    def get_rgba(self, copy: bool = True) -> np.ndarray:
        """uint8 [H,W,4], alpha 255 — what `cam.get_rgba()` returns (simple_env.py:1380; generate_images.py:428).  Render, pack and
        the copy into a PINNED host buffer are one stream-ordered sequence with a single wait (Renderer.render_rgba8_host).

        copy=True (default): a fresh array per call (a 640x480x4 memcpy, ~50 us) — safe for a stereo pair, several cameras on one
        renderer, or a caller that keeps its last N observations.  copy=False: a VIEW of the renderer's pinned ring of depth 2, which all
        cameras of one resolution on this renderer SHARE — the third get_rgba() at that resolution, from any of them, overwrites the
        first; for the reference's own callers, which copy at once (generate_images.py:431 `.copy()`, simple_env.py:1386 `.astype`)."""
        img = self._r.render_rgba8_host(self._camera(), self._scene, config=self._config)
        return img.copy() if copy else img

    def _rgb_depth(self):
        """(rgb [H,W,3] float32 on the GPU, depth [H,W] float32 on the GPU): depth = the scene's expected view depth along
        the optical axis, sum(T alpha z) / coverage — the Gaussian-scene counterpart of Isaac Sim's
        distance_to_image_plane (which the reference reads off the COLLISION MESH, simple_env.py:1401-1409) — and +inf
        where the pixel's coverage is below 1e-4 (nothing hit: what the simulator reports there)."""
        import torch
        rgb, aux = self._r.render(self._camera(), self._scene, config=self._config, return_aux=True)
        cov = aux[..., 1]
        depth = torch.where(cov > 1.0e-4, aux[..., 0] / cov.clamp_min(1.0e-4), torch.full_like(cov, float("inf")))
        return rgb, depth

    def get_current_frame(self) -> dict:
        """{'rgba': uint8 [H,W,4], 'distance_to_image_plane': float32 [H,W]} (simple_env.py:286,1425,1659)."""
        rgb, depth = self._rgb_depth()
        return {"rgba": self._r.pack_rgba8(rgb).cpu().numpy(), "distance_to_image_plane": depth.cpu().numpy()}

    def get_depth(self, clip=(0.1, 6.5)) -> np.ndarray:
        """float32 [H,W] metres, limited to [0.1, 6.5] exactly as SimpleVLNEnv.get_depth does with the simulator's
        distance_to_image_plane (simple_env.py:1573-1578: astype(float32), np.clip(depth, 0.1, 6.5)); a pixel that hit
        nothing (inf) therefore reads 6.5."""
        _, depth = self._rgb_depth()
        d = depth.cpu().numpy().astype(np.float32)
        return np.clip(d, clip[0], clip[1]) if clip is not None else d

    def add_distance_to_image_plane_to_frame(self):        # simple_env.py:850 — always available here
        return None


_NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def _prim_body(text: str, header_re: str):
    """The `{ ... }` body of the first prim whose header matches, or None (brace matching; no nested parsing needed for
    the flat stanzas the reference's stages hold)."""
    m = re.search(header_re, text, re.S)
    if not m:
        return None
    i = text.find("{", m.end())
    if i < 0:
        return None
    depth, j = 0, i
    while j < len(text):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1
    return None


def _vec3(body: str, name: str):
    m = re.search(r"\b(?:double3|float3|half3)\s+" + re.escape(name) + r"\s*=\s*\(\s*(" + _NUM + r")\s*,\s*(" + _NUM + r")\s*,\s*(" + _NUM + r")\s*\)", body)
    return tuple(float(m.group(k)) for k in (1, 2, 3)) if m else None


def parse_scene_usda(text: str) -> dict:
    """What a SAGE-3D scene stage (`{scene_id}.usda`, produced by `sage3d_usda_builder.build_usda_content` :93-149 from
    Data/template.usda) says about the Gaussian asset: the USDZ path referenced by /World/gauss
    (`prepend references = @...usdz[gauss.usda]@`, template.usda:115-117), the collision payload of
    /World/scene_collision (:156-158), and the gauss prim's transform ops — `double3 xformOp:rotateXYZ / :scale /
    :translate` and their `xformOpOrder` (:119-123) — plus the stage's upAxis and metersPerUnit (:105-108)."""
    out = {"usdz": None, "collision": None, "rotate_xyz": None, "scale": None, "translate": None, "xform_op_order": None,
           "up_axis": None, "meters_per_unit": None}
    m = re.search(r"@([^@\n]+\.usdz)\[gauss\.usda\]@", text)
    if m:
        out["usdz"] = m.group(1)
    m = re.search(r"prepend\s+payload\s*=\s*@([^@\n]+)@", text)
    if m is None:
        m = re.search(r"@([^@\n]+_collision\.usd[ac]?)@", text)
    if m:
        out["collision"] = m.group(1)
    body = _prim_body(text, r'over\s+"gauss"\s*(?:\([^)]*\))?')
    if body is not None:
        out["rotate_xyz"] = _vec3(body, "xformOp:rotateXYZ")
        out["scale"] = _vec3(body, "xformOp:scale")
        out["translate"] = _vec3(body, "xformOp:translate")
        m = re.search(r"xformOpOrder\s*=\s*\[([^\]]*)\]", body)
        if m:
            out["xform_op_order"] = tuple(t.strip().strip('"') for t in m.group(1).split(",") if t.strip())
    m = re.search(r'upAxis\s*=\s*"([XYZ])"', text)
    if m:
        out["up_axis"] = m.group(1)
    m = re.search(r"metersPerUnit\s*=\s*(" + _NUM + ")", text)
    if m:
        out["meters_per_unit"] = float(m.group(1))
    return out


def asset_model_to_world(parsed: dict) -> np.ndarray:
    """The 4x4 model->world matrix of the gauss prim from `parse_scene_usda`'s ops, applied in `xformOpOrder`
    (USD: the first op listed is the outermost; rotateXYZ = Rz Ry Rx for column vectors).  The template's
    (translate 0, rotateXYZ (-90,0,0), scale 1) gives `scenes.MODEL_TO_WORLD`.  The renderer takes rigid transforms
    only (the view must stay rigid): a scale other than 1 is refused here rather than silently mis-culled — rescale the
    Gaussians' means and scales instead."""
    import math
    order = parsed.get("xform_op_order") or ("xformOp:translate", "xformOp:rotateXYZ", "xformOp:scale")
    M = np.eye(4)
    for op in order:
        T = np.eye(4)
        if op == "xformOp:translate":
            T[:3, 3] = parsed.get("translate") or (0.0, 0.0, 0.0)
        elif op == "xformOp:rotateXYZ":
            rx, ry, rz = (math.radians(v) for v in (parsed.get("rotate_xyz") or (0.0, 0.0, 0.0)))
            cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
            Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            T[:3, :3] = Rz @ Ry @ Rx
        elif op == "xformOp:scale":
            sc = parsed.get("scale") or (1.0, 1.0, 1.0)
            if any(abs(v - 1.0) > 1e-9 for v in sc):
                raise ValueError(f"xformOp:scale = {sc}: only rigid asset transforms are supported")
        else:
            raise ValueError(f"unsupported xform op {op!r}")
        M = M @ T
    return M
