"""sage_gs — MI355X-native 3D Gaussian Splatting scene renderer behind `render(camera, gaussians)`.

Stands in for the Isaac Sim render step of Galery23/SAGE-3D_Official.  The
forward path (SH -> EWA projection -> AABB -> 16x16 tile binning -> per-tile radix depth sort ->
front-to-back composite) is hand-written HIP for gfx950 in ``csrc/``, reached through the C ABI of
``include/sage_gs.h``; this package is the thin Python host side.
"""
import os as _os

# The library overlaps independent frames on three streams of its own ("lanes") next to the caller's.  The ROCm runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two lanes that land on one queue serialise:
# measured 3070 vs 4100 frames/s on the same box (profiles/r03y_*).  Eight queues keep the lanes, the caller's stream and an
# exchange stream apart.  Read by the runtime when it initialises (the first HIP call of the process): set here, before the
# renderer makes one; an application that initialises HIP earlier sets it itself.
# A process-wide side effect (torch's HIP runtime reads the same variable), so: never overriding a value the application set,
# switched off by SAGE_GS_KEEP_ENV=1, and recorded in HW_QUEUES_SET_BY_PACKAGE / logged on the "sage_gs" logger.
HW_QUEUES_SET_BY_PACKAGE = False
if "GPU_MAX_HW_QUEUES" not in _os.environ and _os.environ.get("SAGE_GS_KEEP_ENV") != "1":
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    HW_QUEUES_SET_BY_PACKAGE = True
    import logging as _logging
    _logging.getLogger("sage_gs").info("GPU_MAX_HW_QUEUES=8 set for this process (SAGE_GS_KEEP_ENV=1 leaves the environment alone)")

from . import _capi  # noqa: F401,E402
from ._capi import SgsError  # noqa: F401,E402

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
