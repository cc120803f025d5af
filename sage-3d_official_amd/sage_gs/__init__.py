"""sage_gs — MI355X-native 3D Gaussian Splatting scene renderer behind `render(camera, gaussians)`.

Stands in for the Isaac Sim render step of Galery23/SAGE-3D_Official.  The
forward path (SH -> EWA projection -> AABB -> 16x16 tile binning -> per-tile radix depth sort ->
front-to-back composite) is hand-written HIP for gfx950 in ``csrc/``, reached through the C ABI of
``include/sage_gs.h``; this package is the thin Python host side.
"""
from . import _capi  # noqa: F401
from ._capi import SgsError  # noqa: F401

__all__ = ["Camera", "RenderConfig", "Gaussians", "Renderer", "Scene", "render", "default_renderer",
           "SgsError", "scenes"]


def __getattr__(name):          # torch is imported only when the renderer is actually used
    if name in ("Camera", "RenderConfig", "Gaussians", "Renderer", "Scene", "render", "default_renderer"):
        from . import renderer
        return getattr(renderer, name)
    if name in ("scenes", "camera", "dist", "adapter", "ply"):
        import importlib
        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
