"""ctypes binding of oracle/sgs_oracle.c (the C restatement).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED — see the header of sgs_oracle.c / oracle_np.py.  Loaded only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("view", C.c_float * 16)]


class OrcConfig(C.Structure):
    _fields_ = [("near_z", C.c_float), ("far_z", C.c_float), ("dilation", C.c_float),
                ("clamp", C.c_float), ("alpha_min", C.c_float), ("alpha_max", C.c_float),
                ("t_min", C.c_float), ("bg", C.c_float * 3), ("sh_degree", C.c_int32)]


class OrcFrame(C.Structure):
    _fields_ = [("N", C.c_int64), ("n_visible", C.c_int64), ("D", C.c_int64), ("D_f", C.c_int64),
                ("gx", C.c_int32), ("gy", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("depth_bits", C.POINTER(C.c_uint32)), ("rect", C.POINTER(C.c_int32)),
                ("tiles", C.POINTER(C.c_int32)), ("xy", C.POINTER(C.c_float)),
                ("conic", C.POINTER(C.c_float)), ("opacity", C.POINTER(C.c_float)),
                ("rgb", C.POINTER(C.c_float)), ("offsets", C.POINTER(C.c_int64)),
                ("ids", C.POINTER(C.c_int32)), ("consumed", C.POINTER(C.c_int64)),
                ("image", C.POINTER(C.c_float)), ("final_T", C.POINTER(C.c_float)),
                ("n_contrib", C.POINTER(C.c_int32)), ("width", C.c_int32), ("height", C.c_int32),
                ("depth_img", C.POINTER(C.c_float)), ("margin", C.POINTER(C.c_float))]


def build(force=False):
    """Compile liborc_f64.so / liborc_f32.so in place (gcc, a few seconds)."""
    need = force or any(not os.path.exists(os.path.join(_HERE, n)) or
                        os.path.getmtime(os.path.join(_HERE, n)) < os.path.getmtime(os.path.join(_HERE, "sgs_oracle.c"))
                        for n in ("liborc_f64.so", "liborc_f32.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


_libs = {}


def _lib(real="f64"):
    if real not in _libs:
        # ORC_LIB_SUFFIX=_asan: the sanitizer build of the checker (oracle/Makefile `asan`; scripts/oracle_asan.sh)
        path = os.path.join(_HERE, f"liborc_{real}{os.environ.get('ORC_LIB_SUFFIX', '') if real == 'f64' else ''}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.orc_render.restype = C.POINTER(OrcFrame)
        lib.orc_render.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(OrcCamera), C.POINTER(OrcConfig),
                                   C.c_int, C.c_int, C.c_int]
        lib.orc_frame_free.argtypes = [C.POINTER(OrcFrame)]
        lib.orc_frame_free.restype = None
        lib.orc_max_threads.restype = C.c_int
        lib.orc_pixel_variants.restype = C.c_int64
        lib.orc_pixel_variants.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcConfig), C.c_int, C.c_int, C.c_double,
                                           C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        _libs[real] = lib
    return _libs[real]


def max_threads():
    return int(_lib().orc_max_threads())


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


def make_camera(width, height, fx, fy, cx, cy, view):
    cam = OrcCamera(int(width), int(height), float(fx), float(fy), float(cx), float(cy))
    v = np.asarray(view, np.float32).reshape(16)
    for i in range(16):
        cam.view[i] = float(v[i])
    return cam


def make_config(near=0.2, far=1e30, dilation=0.3, clamp=1.3, alpha_min=1.0 / 255.0, alpha_max=0.99,
                t_min=1e-4, background=(0.0, 0.0, 0.0), sh_degree=-1):
    cfg = OrcConfig(near, far, dilation, clamp, alpha_min, alpha_max, t_min)
    for i in range(3):
        cfg.bg[i] = float(background[i])
    cfg.sh_degree = int(sh_degree)
    return cfg


class Recheck:
    """The oracle's two-sided answer for threshold-sensitive pixels (orc_pixel_variants): holds the C frame of one
    oracle run alive.  recheck(ys, xs, got) -> (best_err[K], leaves[K], capped[K]): for each listed pixel the smallest
    max-over-channels |variant - got| over every admissible set of alpha-cut / stop decisions inside the margin."""

    def __init__(self, lib, fp, cfg, rel_margin, cap=12):
        self._lib, self._fp, self._cfg, self.rel_margin, self.cap = lib, fp, cfg, float(rel_margin), int(cap)

    def __call__(self, ys, xs, got):
        got = np.ascontiguousarray(got, np.float32).reshape(-1, 3)
        best = np.zeros(len(got)); leaves = np.zeros(len(got), np.int64); capped = np.zeros(len(got), bool)
        e, c = C.c_double(), C.c_int()
        for i, (y, x) in enumerate(zip(ys, xs)):
            g = (C.c_float * 3)(*[float(v) for v in got[i]])
            leaves[i] = self._lib.orc_pixel_variants(self._fp, C.byref(self._cfg), int(x), int(y), self.rel_margin,
                                                     self.cap, g, C.byref(e), C.byref(c))
            best[i], capped[i] = e.value, bool(c.value)
        return best, leaves, capped

    def close(self):
        if self._fp is not None:
            self._lib.orc_frame_free(self._fp)
            self._fp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


REL_MARGIN = 1.0e-4     # relative distance to a threshold below which a differently rounded evaluation may decide otherwise


def render(means, scales, quats, opacities, sh, sh_degree, cam, cfg=None, tile_row_begin=0,
           tile_row_end=-1, threads=0, real="f64", want="all"):
    """Run the C oracle.  `cam` is an oracle_np.Camera-like object or an OrcCamera.

    Returns (image float32 [H,W,3], aux dict).  want="image" skips copying the intermediates.
    aux["recheck"] (checker build only) evaluates threshold-sensitive pixels two-sidedly — see Recheck.
    """
    if not isinstance(cam, OrcCamera):
        cam = make_camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.view)
    if cfg is None:
        cfg = make_config()
    elif not isinstance(cfg, OrcConfig):
        cfg = make_config(cfg.near, cfg.far, cfg.dilation, cfg.clamp, cfg.alpha_min, cfg.alpha_max,
                          cfg.t_min, cfg.background, cfg.sh_degree)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (means, scales, quats, opacities, sh)]
    N = arrs[0].shape[0]
    lib = _lib(real)
    fp = lib.orc_render(N, int(sh_degree), *[a.ctypes.data for a in arrs], C.byref(cam), C.byref(cfg),
                        int(tile_row_begin), int(tile_row_end), int(threads))
    if not fp:
        raise MemoryError("orc_render failed")
    f = fp.contents
    H, W = f.height, f.width
    img = _np(f.image, (H, W, 3), np.float32)
    aux = dict(n_visible=int(f.n_visible), D=int(f.D), D_f=int(f.D_f), gx=int(f.gx), gy=int(f.gy))
    if want == "all":
        T = f.gx * f.gy
        aux.update(
            depth_bits=_np(f.depth_bits, (N,), np.uint32), rect=_np(f.rect, (N, 4), np.int32),
            tiles=_np(f.tiles, (N,), np.int32), xy=_np(f.xy, (N, 2), np.float32),
            conic=_np(f.conic, (N, 3), np.float32), opacity=_np(f.opacity, (N,), np.float32),
            rgb=_np(f.rgb, (N, 3), np.float32), offsets=_np(f.offsets, (T + 1,), np.int64),
            ids=_np(f.ids, (int(f.D),), np.int32), consumed=_np(f.consumed, (T,), np.int64),
            final_T=_np(f.final_T, (H, W), np.float32), n_contrib=_np(f.n_contrib, (H, W), np.int32))
    aux["margin"] = _np(f.margin, (H, W), np.float32)
    aux["depth_image"] = _np(f.depth_img, (H, W), np.float32)
    aux["final_T"] = _np(f.final_T, (H, W), np.float32)
    if real == "f64":
        aux["recheck"] = Recheck(lib, fp, cfg, REL_MARGIN)      # owns the frame from here on
    else:
        lib.orc_frame_free(fp)
    return img, aux
