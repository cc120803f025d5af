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

    def get_rgba(self) -> np.ndarray:
        """uint8 [H,W,4], alpha 255 — what `cam.get_rgba()` returns (simple_env.py:1380; generate_images.py:428)."""
        return self._r.pack_rgba8(self.get_rgb_tensor()).cpu().numpy()

    def get_current_frame(self) -> dict:
        """{'rgba': uint8 [H,W,4], 'distance_to_image_plane': float32 [H,W]} (simple_env.py:286,1425,1659):
        the depth entry is the Gaussian scene's expected view depth (f-4), 0 where nothing was hit."""
        rgb, aux = self._r.render(self._camera(), self._scene, config=self._config, return_aux=True)
        return {"rgba": self._r.pack_rgba8(rgb).cpu().numpy(),
                "distance_to_image_plane": aux[..., 0].cpu().numpy()}

    def add_distance_to_image_plane_to_frame(self):        # simple_env.py:850 — always available here
        return None


def parse_scene_usda(text: str) -> dict:
    """What `sage3d_usda_builder.build_usda_content` (:93-149) substitutes into Data/template.usda: the USDZ
    asset path of /World/gauss, the collision USD path and the gauss prim's rotateXYZ (template.usda:115-124)."""
    out = {"usdz": None, "collision": None, "rotate_xyz": None}
    m = re.search(r"@([^@\n]+\.usdz)\[gauss\.usda\]@", text)
    if m:
        out["usdz"] = m.group(1)
    m = re.search(r"@([^@\n]+_collision\.usd[ac]?)@", text)
    if m:
        out["collision"] = m.group(1)
    m = re.search(r'over\s+"gauss".*?rotateXYZ\s*=\s*\(([^)]*)\)', text, re.S)
    if m:
        out["rotate_xyz"] = tuple(float(v) for v in m.group(1).split(","))
    return out
