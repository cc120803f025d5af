"""`render(camera, gaussians)` — the Python call surface of the MI355X 3DGS scene renderer.

Host code stays Python on PyTorch-ROCm (device memory, streams); every frame is one call into the
C ABI of ``libsage_gs.so`` (include/sage_gs.h) with raw device pointers.  This module stands where the
reference's Isaac Sim render step stood:

    reference                                             here
    ----------------------------------------------------  ---------------------------------------
    open_stage(usd)  (simple_env.py:219)                   Renderer.upload(gaussians) -> Scene
    cam.set_world_pose(position, orientation) (:1284)      Camera(view=...) / camera.from_isaac_pose
    world.step(render=True)x2 ; cam.get_rgba() (:1368-80)  Renderer.render(camera, scene)
    frame loop (generate_images.py:408-436)                Renderer.render_batch(cameras, scene)

There is no CPU fallback: importing this module needs the built HIP library, rendering needs a GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi


@dataclass
class Camera:
    """Pinhole camera, +Z forward / +X right / +Y down.  `view` is world->camera (4x4, rigid)."""
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    view: np.ndarray = field(default_factory=lambda: np.eye(4))

    @property
    def tile_rows(self) -> int:
        return (self.height + 15) // 16

    @property
    def tile_cols(self) -> int:
        return (self.width + 15) // 16


@dataclass
class RenderConfig:
    """Constants of stages S2-S6; defaults are the canonical values."""
    near: float = 0.2
    far: float = 1.0e30
    dilation: float = 0.3
    clamp: float = 1.3
    alpha_min: float = 1.0 / 255.0
    alpha_max: float = 0.99
    t_min: float = 1.0e-4
    background: Sequence[float] = (0.0, 0.0, 0.0)
    sh_degree: int = -1


@dataclass
class Gaussians:
    """A 3DGS scene with activations applied: linear scales, (w,x,y,z) quaternions,
    opacities in (0,1), SH coefficients [N,(d+1)^2,3].  `model_to_world` is the asset transform of
    Data/template.usda:115-124 (rotateXYZ -90,0,0 for SAGE-3D scenes); it is applied by moving the
    camera into model space, which is exact for rigid transforms."""
    means: torch.Tensor
    scales: torch.Tensor
    quats: torch.Tensor
    opacities: torch.Tensor
    sh: torch.Tensor
    sh_degree: int
    model_to_world: Optional[np.ndarray] = None

    def __len__(self):
        return int(self.means.shape[0])


class Scene:
    """Device-resident, re-laid-out copy of a Gaussians object (wave-chunked float4 rows)."""

    def __init__(self, renderer: "Renderer", handle, n, sh_degree, model_to_world):
        self._r, self.handle, self.n, self.sh_degree = renderer, handle, n, sh_degree
        self.model_to_world = model_to_world            # (validated by Renderer.upload* BEFORE the upload: _check_model_to_world)

    def free(self):
        if self.handle:
            self._r._lib.sgs_scene_free(self._r._ctx, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _rigid(views: np.ndarray) -> np.ndarray:
    """[...,4,4] float64 CAMERA poses (world -> camera) -> the same, with every 3x3 that is measurably off orthonormal AFTER the cast to
    fp32 (a view composed or inverted in fp32: torch.linalg.inv of a c2w matrix, poses parsed from 6-digit text) replaced by the nearest
    rotation (polar decomposition).  The C ABI wants the rows orthonormal to 1e-5 (include/sage_gs.h) and says so with an error at
    enqueue time; a Python caller gets the projection instead.  Views already orthonormal to 2e-6 pass through bit for bit, and
    one that is off by more than 1e-3 is not a pose with rounding noise: it is left alone and the library reports it.
    Applied to the camera's pose ONLY, before it is composed with a scene's model_to_world — an asset transform is validated on its own
    (_check_model_to_world): a USD xformOp:scale of 1.0003 is a real scale, not rounding noise, and must not be projected away."""
    v = np.array(views, np.float64, copy=True)
    flat = v.reshape(-1, 4, 4)
    if flat.shape[0] == 1:
        # one pose — once per get_rgba() of the adapter: the same test in scalar arithmetic on the fp32-rounded entries (the small-array
        # NumPy form below costs 15 us of a 0.3-ms call); the rare pose that needs projecting takes the general path
        (a, b, c), (d, e, f), (g, h, i) = flat[0, :3, :3].astype(np.float32).tolist()
        dev = max(abs(a * a + b * b + c * c - 1.0), abs(d * d + e * e + f * f - 1.0), abs(g * g + h * h + i * i - 1.0),
                  abs(a * d + b * e + c * f), abs(a * g + b * h + c * i), abs(d * g + e * h + f * i))
        if not (1.5e-6 < dev < 1.1e-3):              # (clear of both thresholds: the verdict is the loop's)
            return v
    for m in flat:
        r = m[:3, :3].astype(np.float32).astype(np.float64)
        dev = np.abs(r @ r.T - np.eye(3)).max()
        if 2.0e-6 < dev < 1.0e-3:
            u, _, vt = np.linalg.svd(m[:3, :3])
            m[:3, :3] = u @ vt
    return v


def _check_model_to_world(m) -> np.ndarray:
    """A scene's model -> world transform must be RIGID (rotation + translation): the renderer applies it by moving the camera into model
    space, which is exact only then.  Off orthonormal by up to 2e-6 (a rotation written in fp32): accepted as it is.  Up to 1e-5 — what the
    library itself tolerates; a product of two fp32 rotations lands at 1e-6 .. 5e-6 — the 3x3 is replaced by the nearest rotation (polar
    decomposition): rounding noise, not a transform.  Anything more — a scale (a USD xformOp:scale of 1.0003 is a REAL scale: 6e-4), a shear,
    a mirror — raises here, with the fix, instead of being silently re-orthonormalised (round 4) or rejected later by the library's
    per-frame rigidity check with a message about the camera.  Called BEFORE the scene is uploaded: a refusal leaves nothing on the device."""
    m = np.array(m, np.float64, copy=True).reshape(4, 4)
    r = m[:3, :3]
    dev = float(np.abs(r @ r.T - np.eye(3)).max())
    if not (dev <= 1.0e-5) or not np.allclose(m[3], [0.0, 0.0, 0.0, 1.0], atol=1e-12) or np.linalg.det(r) < 0:
        raise ValueError(f"model_to_world is not a rigid transform (its 3x3 is off orthonormal by {dev:.3g}): bake the asset's scale / shear into the "
                         "Gaussians' means and scales (a USD xformOp:scale s multiplies both) and pass the rotation + translation only")
    if dev > 2.0e-6:
        u, _, vt = np.linalg.svd(r)
        m[:3, :3] = u @ vt
    return m


def _as_f32(t: torch.Tensor, device, shape_tail):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t, np.float32))
    t = t.to(device=device, dtype=torch.float32).contiguous()
    if tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"expected [N,{','.join(map(str, shape_tail))}], got {tuple(t.shape)}")
    return t


class Renderer:
    """One rendering context bound to one GPU (one process per GPU is the intended deployment)."""

    def __init__(self, device=None, record_capacity: Optional[int] = None, lib: Optional[_capi.Lib] = None, *,
                 lanes: Optional[int] = None, group: Optional[int] = None, group_lanes: Optional[int] = None,
                 morton: Optional[bool] = None, fine_tile_pixels: Optional[int] = None, fine_tile_growth: Optional[float] = None):
        """lanes / group / group_lanes / morton / record_capacity / fine_tile_pixels: include/sage_gs.h `sgs_tuning` (None = the library's
        default, what the bench runs).  The library reads nothing from the environment; frames do not depend on any of these, bit for bit —
        except fine_tile_pixels / fine_tile_growth (frames of at most that many pixels are rendered through 8x8-pixel tiles, of a quarter of it
        through 4x4, as long as a split multiplies the frame's records by no more than fine_tile_growth: the same splats reach every pixel,
        the blend's coordinates are relative to another tile origin, so frames agree to fp32 rounding)."""
        self._lib = lib or _capi.Lib()
        if not torch.cuda.is_available():
            raise RuntimeError("sage_gs.Renderer needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        ctx = C.c_void_p()
        self._lib.check(self._lib.sgs_create(index, _capi.BACKEND_HIP, C.byref(ctx)))
        self._ctx = ctx
        if any(v is not None for v in (record_capacity, lanes, group, group_lanes, morton, fine_tile_pixels, fine_tile_growth)):
            self.set_tuning(lanes=lanes, group=group, group_lanes=group_lanes, morton=morton, record_capacity=record_capacity,
                            fine_tile_pixels=fine_tile_pixels, fine_tile_growth=fine_tile_growth)
        self.last_stats = None

    def tuning(self) -> dict:
        t = _capi.SgsTuning()
        self._lib.check(self._lib.sgs_get_tuning(self._ctx, C.byref(t)), self._ctx)
        return {k: (float if k == "fine_tile_growth" else int)(getattr(t, k)) for k, _ in t._fields_}

    def set_tuning(self, **kw):
        """sgs_set_tuning: any of lanes, group, group_lanes, morton, record_capacity, fine_tile_pixels (the others keep their values); applies to the scenes
        uploaded and the frames issued afterwards."""
        t = _capi.SgsTuning()
        self._lib.check(self._lib.sgs_get_tuning(self._ctx, C.byref(t)), self._ctx)
        for k, v in kw.items():
            if k not in dict(t._fields_):
                raise TypeError(f"unknown tuning field {k!r}")
            if v is not None:
                setattr(t, k, float(v) if k == "fine_tile_growth" else int(v))
        self._lib.check(self._lib.sgs_set_tuning(self._ctx, C.byref(t)), self._ctx)

    # -- scene ------------------------------------------------------------------------------------
    def upload(self, g: Gaussians) -> Scene:
        m2w = None if g.model_to_world is None else _check_model_to_world(g.model_to_world)      # (raises before anything is on the device)
        n = len(g)
        k = (g.sh_degree + 1) ** 2
        with torch.cuda.device(self.device):
            means = _as_f32(g.means, self.device, (3,))
            scales = _as_f32(g.scales, self.device, (3,))
            quats = _as_f32(g.quats, self.device, (4,))
            opac = _as_f32(g.opacities.reshape(n), self.device, ())
            sh = _as_f32(g.sh.reshape(n, k, 3), self.device, (k, 3))
            torch.cuda.synchronize(self.device)
            h = C.c_void_p()
            self._lib.check(self._lib.sgs_scene_upload(self._ctx, n, int(g.sh_degree), means.data_ptr(),
                                                       scales.data_ptr(), quats.data_ptr(), opac.data_ptr(),
                                                       sh.data_ptr(), 1, C.byref(h)), self._ctx)
        return Scene(self, h, n, int(g.sh_degree), m2w)

    def upload_compressed(self, chunks, packed, sh, sh_degree: int, model_to_world=None, sh_decode: Optional[str] = None) -> Scene:
        """A scene from the PlayCanvas compressed.ply payload (ply.read_compressed_payload: chunks float32 [nch,18], packed uint32 [n,4],
        sh uint8 [n, 3 k_rest] or None): copied to the device as it is — 16 B + SH bytes per Gaussian — and dequantised there by the
        layout kernel (sgs_scene_upload_compressed); the 8-bit SH coefficients stay bytes in HBM and are dequantised by the projection
        kernel every frame.  NumPy arrays or tensors; tensors already on this device are used in place.
        sh_decode: how a coefficient byte becomes a float — "bin_centre", "linear255" or "bin_centre_ends" (ply.decode_sh_bytes,
        include/sage_gs.h SGS_SH_DECODE_*).  REQUIRED at degree > 0, no default: the tool the reference delegates the decode to could not be
        inspected offline, so say what the tool that wrote / would decompress your file uses."""
        m2w = None if model_to_world is None else _check_model_to_world(model_to_world)          # (raises before anything is on the device)

        def dev(a, dt):
            if a is None:
                return None
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            if t.dtype != dt:
                t = t.view(dt) if t.element_size() == torch.empty((), dtype=dt).element_size() else t.to(dt)
            return t.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            c = dev(chunks, torch.float32)
            p = dev(packed if isinstance(packed, torch.Tensor) else np.ascontiguousarray(packed).view(np.int32), torch.int32)
            b = dev(sh, torch.uint8)
            n, nch = int(p.shape[0]), int(c.shape[0])
            if tuple(c.shape) != (nch, 18) or tuple(p.shape) != (n, 4) or nch != (n + 255) // 256:
                raise ValueError("chunks must be [ceil(n/256), 18] and packed [n, 4]")
            k_rest = (int(sh_degree) + 1) ** 2 - 1
            if (k_rest > 0) != (b is not None) or (b is not None and tuple(b.shape) != (n, 3 * k_rest)):
                raise ValueError(f"sh must be uint8 [n, {3 * k_rest}] at degree {sh_degree} (None at degree 0)")
            if k_rest > 0 and sh_decode is None:
                raise ValueError(f"sh_decode is required for a scene with 8-bit SH coefficients: one of {sorted(_capi.SH_DECODE)} (there is no "
                                 "default: ply.decode_sh_bytes / include/sage_gs.h say why)")
            if sh_decode is not None and sh_decode not in _capi.SH_DECODE:
                raise ValueError(f"sh_decode must be one of {sorted(_capi.SH_DECODE)}")
            z = _capi.SgsCompressedScene(n, nch, int(sh_degree), _capi.SH_DECODE[sh_decode] if sh_decode is not None else 0, c.data_ptr(), p.data_ptr(), b.data_ptr() if b is not None else None)
            torch.cuda.synchronize(self.device)
            h = C.c_void_p()
            self._lib.check(self._lib.sgs_scene_upload_compressed(self._ctx, C.byref(z), 1, C.byref(h)), self._ctx)
        return Scene(self, h, n, int(sh_degree), m2w)

    def _scene_of(self, g):
        if isinstance(g, Scene):
            if g._r is not self:
                raise ValueError("this Scene was uploaded by another Renderer (another context, possibly another GPU); "
                                 "upload the Gaussians with the renderer that draws them")
            if g.handle is None:
                raise ValueError("this Scene has been freed")
            return g
        cached = getattr(g, "_sgs_scene", None)
        if cached is None or cached._r is not self or cached.handle is None:
            cached = self.upload(g)
            g._sgs_scene = cached
        return cached

    # -- camera / config marshalling ---------------------------------------------------------------
    @staticmethod
    def _c_camera(cam: Camera, scene: Scene) -> _capi.SgsCamera:
        view = np.asarray(cam.view.detach().cpu().numpy() if isinstance(cam.view, torch.Tensor) else cam.view,
                          np.float64).reshape(4, 4)
        view = _rigid(view)                       # (the camera's pose alone: see _rigid)
        if scene.model_to_world is not None:
            view = view @ scene.model_to_world.reshape(4, 4)
        return _capi.make_camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                                 view.astype(np.float32).tolist())

    _CAM_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
                           ("view", "<f4", (16,))])

    @classmethod
    def _c_cameras(cls, cameras: Sequence[Camera], scene: Scene) -> np.ndarray:
        """The sgs_camera array of a batch, marshalled in one go (a structured array with sgs_camera's layout): per-camera
        ctypes marshalling costs ~30 us, as much as a light band of tile rows takes to render."""
        assert cls._CAM_DTYPE.itemsize == C.sizeof(_capi.SgsCamera)
        b = len(cameras)
        views = np.stack([np.asarray(c.view.detach().cpu().numpy() if isinstance(c.view, torch.Tensor) else c.view,
                                     np.float64).reshape(4, 4) for c in cameras])
        views = _rigid(views)                     # (the cameras' poses alone: see _rigid)
        if scene.model_to_world is not None:
            views = views @ scene.model_to_world.reshape(4, 4)
        arr = np.zeros(b, cls._CAM_DTYPE)
        arr["width"] = [c.width for c in cameras]; arr["height"] = [c.height for c in cameras]
        arr["fx"] = [c.fx for c in cameras]; arr["fy"] = [c.fy for c in cameras]
        arr["cx"] = [c.cx for c in cameras]; arr["cy"] = [c.cy for c in cameras]
        arr["view"] = views.reshape(b, 16).astype(np.float32)
        return arr

    def _c_config(self, cfg: Optional[RenderConfig], flags=0) -> _capi.SgsConfig:
        k = self._lib.default_config()
        if cfg is not None:
            k.near_z, k.far_z, k.dilation, k.clamp = cfg.near, cfg.far, cfg.dilation, cfg.clamp
            k.alpha_min, k.alpha_max, k.t_min = cfg.alpha_min, cfg.alpha_max, cfg.t_min
            for i in range(3):
                k.bg[i] = float(cfg.background[i])
            k.sh_degree = int(cfg.sh_degree)
        k.flags = flags
        return k

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- frames -----------------------------------------------------------------------------------
    def render(self, camera: Camera, gaussians, *, config: Optional[RenderConfig] = None,
               out: Optional[torch.Tensor] = None, out_band: Optional[torch.Tensor] = None,
               tile_rows=None, timing=False, sync=True, full_sort=False, out_aux: Optional[torch.Tensor] = None,
               return_aux=False, pipelined=False, loose_cull=False, interleave=None, chunk_cull=True, stats=False,
               deep_cull=True, fine_tiles=True):
        """One frame -> float32 tensor [H,W,3] on this renderer's device (linear RGB).

        fine_tiles=False (tests, A/B): SGS_FLAG_NO_FINE_TILES — 16x16-pixel tiles whatever the frame's size (by default a frame of at most
        `fine_tile_pixels` pixels, 640x480, goes through 8x8-pixel tiles and one of a quarter of that through 4x4: DESIGN.md §4.8).

        stats=True also counts D_f (records consumed by the composite; SGS_FLAG_STATS) — bookkeeping that costs a sweep ~4 %,
        so it is opt-in; N_v, D and the stage times (timing=True) are always available.

        tile_rows=(r0,r1) renders only that band of 16-pixel tile rows (multi-GPU sharding); other rows
        of `out` are left untouched.  With `out_band` (a [>=band rows, W, 3] slab) only the band is
        stored, at the top of the slab, and the slab is returned.  sync=False enqueues on the current
        stream without waiting (collect with .sync()).  pipelined=True (with sync=False) lets the frame overlap
        with other pipelined frames on the library's internal streams: for sweeps of independent frames; the
        output is complete only after .sync().

        interleave=(stride, phase) renders the tile rows phase, phase+stride, ... of the frame (the balanced way to
        shard one frame over `stride` GPUs) into `out_band`, a COMPACT [>= 16 * owned rows, W, 3] image: owned row k
        (frame tile row k*stride + phase) is stored at pixel rows [16k, 16k+16).  tile_rows then indexes owned rows.

        deep_cull=False (tests, A/B): SGS_FLAG_NO_DEEP — no window of a long-lived tile is culled against its live pixels before it is
        ranked; the frame must not change."""
        scene = self._scene_of(gaussians)
        r0, r1 = (0, -1) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
        stride, phase = (1, 0) if interleave is None else (int(interleave[0]), int(interleave[1]))
        if stride > 1:
            if out_band is None or out is not None or return_aux or out_aux is not None:
                raise ValueError("interleave renders into out_band (a compact image of the owned rows) only")
            if not 0 <= phase < stride:
                raise ValueError(f"interleave phase {phase} outside [0, {stride})")
            owned = len(range(phase, camera.tile_rows, stride))
            need = 16 * (owned if r1 < 0 else min(r1, owned))
            if (out_band.device != self.device or out_band.dtype != torch.float32 or not out_band.is_contiguous()
                    or out_band.dim() != 3 or out_band.shape[0] < need or tuple(out_band.shape[1:]) != (camera.width, 3)):
                raise ValueError("out_band must be a contiguous float32 [>= 16 * owned rows, W, 3] tensor on the device")
            ptr, ret = out_band.data_ptr(), out_band
        elif out_band is not None:
            if tile_rows is None:
                raise ValueError("out_band needs tile_rows")
            if r1 < 0 or r1 > camera.tile_rows:      # "to the end", as the C ABI reads it
                r1 = camera.tile_rows
            if not 0 <= r0 <= r1:
                raise ValueError(f"tile_rows {tile_rows} is not a band of the frame's {camera.tile_rows} tile rows")
            y0, y1 = r0 * 16, min(r1 * 16, camera.height)
            if (out_band.device != self.device or out_band.dtype != torch.float32 or not out_band.is_contiguous()
                    or out_band.dim() != 3 or out_band.shape[0] < y1 - y0
                    or tuple(out_band.shape[1:]) != (camera.width, 3)):
                raise ValueError("out_band must be a contiguous float32 [>=band rows, W, 3] tensor on the device")
            # the ABI takes the address of pixel (0,0); only the band's rows are ever dereferenced
            ptr, ret = out_band.data_ptr() - y0 * camera.width * 3 * 4, out_band
        else:
            if out is None:
                out = torch.zeros((camera.height, camera.width, 3), dtype=torch.float32, device=self.device)
            elif (out.device != self.device or out.dtype != torch.float32 or not out.is_contiguous()
                  or tuple(out.shape) != (camera.height, camera.width, 3)):
                raise ValueError("out must be a contiguous float32 [H,W,3] tensor on the renderer's device")
            ptr, ret = out.data_ptr(), out
        flags = (0 if sync else _capi.FLAG_ASYNC) | (_capi.FLAG_TIMING if timing else 0) | \
                (_capi.FLAG_FULL_SORT if full_sort else 0) | \
                (_capi.FLAG_LOOSE_CULL if loose_cull else 0) | \
                (0 if chunk_cull else _capi.FLAG_NO_CHUNK_CULL) | (_capi.FLAG_STATS if stats else 0) | \
                (0 if deep_cull else _capi.FLAG_NO_DEEP) | (0 if fine_tiles else _capi.FLAG_NO_FINE_TILES) | \
                (_capi.FLAG_PIPELINED if (pipelined and not sync) else 0)   # full_sort: test hook, orders every queue completely
        cam, cfg, st = self._c_camera(camera, scene), self._c_config(config, flags), _capi.SgsStats()
        cfg.tile_row_stride, cfg.tile_row_phase = stride, phase
        if return_aux or out_aux is not None:
            # f-4: [H,W,2] = expected view depth sum(T alpha z), coverage 1 - T_final (full-frame buffers only)
            if out_band is not None:
                raise ValueError("aux output is not available together with out_band")
            if out_aux is None:
                out_aux = torch.zeros((camera.height, camera.width, 2), dtype=torch.float32, device=self.device)
            self._lib.check(self._lib.sgs_render_rgbd(self._ctx, scene.handle, C.byref(cam), C.byref(cfg), r0, r1, ptr,
                                                      out_aux.data_ptr(), C.byref(st), self._stream()), self._ctx)
            self.last_stats = st.as_dict() if sync else None
            return ret, out_aux
        self._lib.check(self._lib.sgs_render(self._ctx, scene.handle, C.byref(cam), C.byref(cfg), r0, r1,
                                             ptr, C.byref(st), self._stream()), self._ctx)
        self.last_stats = st.as_dict() if sync else None
        return ret

    def sync(self):
        """Complete frames issued with sync=False; returns the statistics of the last one."""
        st = _capi.SgsStats()
        self._lib.check(self._lib.sgs_frame_sync(self._ctx, C.byref(st)), self._ctx)
        self.last_stats = st.as_dict()
        return self.last_stats

    def row_records(self, n_rows: int, reset: bool = True) -> np.ndarray:
        """Records queued per frame tile row, summed over the frames rendered since the last reset (int64 [n_rows]):
        the per-row cost that cost-balanced tile-row bands are cut from (sage_gs.dist).  Covers completed frames (call after sync())."""
        out = np.zeros(int(n_rows), np.int64)
        self._lib.check(self._lib.sgs_row_records(self._ctx, out.ctypes.data, int(n_rows), 1 if reset else 0), self._ctx)
        return out

    def render_batch(self, cameras: Sequence[Camera], gaussians, *, config: Optional[RenderConfig] = None,
                     out: Optional[torch.Tensor] = None, tile_rows=None, want_stats=False, stats=False,
                     out_bands: Optional[torch.Tensor] = None, interleave=None, fine_tiles=True):
        """B frames of one scene in ONE call into the library (camera-sweep batch): the frames go through the pipelined
        lanes, and the per-frame host work (stream fork, status clear / copy, completion) is paid once per batch.

        out: [B,H,W,3] (allocated when omitted).  tile_rows=(r0,r1) renders only that band of every frame.  With
        `out_bands` — a [B, >= band rows, W, 3] tensor whose frames may be strided views, e.g. the slabs of a sharded
        sweep — only the band is stored, at the top of each slab.  interleave=(stride, phase) with out_bands renders the
        owned tile rows of every frame into compact images (see render()).  want_stats=True also returns every frame's
        statistics (N_v, D, ...); D_f is in them only with stats=True (see render())."""
        scene = self._scene_of(gaussians)
        b = len(cameras)
        if b == 0:
            raise ValueError("no cameras")
        h, w = cameras[0].height, cameras[0].width
        if any(c.height != h or c.width != w for c in cameras):
            raise ValueError("all cameras of a batch must share a resolution")
        r0, r1 = (0, -1) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
        cfg = self._c_config(config, (_capi.FLAG_STATS if stats else 0) | (0 if fine_tiles else _capi.FLAG_NO_FINE_TILES))     # (D_f is counted on request only: stats=True)
        if out_bands is not None:
            if out is not None:
                raise ValueError("give out or out_bands, not both")
            stride, phase = (1, 0) if interleave is None else (int(interleave[0]), int(interleave[1]))
            if stride > 1:
                if not 0 <= phase < stride:
                    raise ValueError(f"interleave phase {phase} outside [0, {stride})")
                owned = len(range(phase, cameras[0].tile_rows, stride))
                need, y0 = 16 * (owned if r1 < 0 else min(r1, owned)), 0
                cfg.tile_row_stride, cfg.tile_row_phase = stride, phase
            else:
                if tile_rows is None:
                    raise ValueError("out_bands needs tile_rows (or interleave)")
                if r1 < 0 or r1 > cameras[0].tile_rows:
                    r1 = cameras[0].tile_rows
                if not 0 <= r0 <= r1:
                    raise ValueError(f"tile_rows {tile_rows} is not a band of the frame's {cameras[0].tile_rows} tile rows")
                y0 = r0 * 16
                need = min(r1 * 16, h) - y0
            t = out_bands
            if (t.device != self.device or t.dtype != torch.float32 or t.dim() != 4 or t.shape[0] < b or t.shape[1] < need
                    or tuple(t.shape[2:]) != (w, 3) or t.stride(3) != 1 or t.stride(2) != 3 or t.stride(1) != 3 * w):
                raise ValueError("out_bands must be float32 [>= B, >= band rows, W, 3] on the device with contiguous frames")
            ptr, frame_stride, ret = t.data_ptr() - y0 * w * 3 * 4, int(t.stride(0)), t
        else:
            if interleave is not None:
                raise ValueError("interleave renders into out_bands only")
            if out is None:
                out = torch.zeros((b, h, w, 3), dtype=torch.float32, device=self.device)
            elif (out.device != self.device or out.dtype != torch.float32 or out.dim() != 4 or out.shape[0] < b
                  or tuple(out.shape[1:]) != (h, w, 3) or not out[0].is_contiguous()):
                raise ValueError("out must be float32 [>= B, H, W, 3] on the renderer's device")
            ptr, frame_stride, ret = out.data_ptr(), int(out.stride(0)), out
        arr = self._c_cameras(cameras, scene)
        stats = (_capi.SgsStats * b)() if want_stats else None
        self._lib.check(self._lib.sgs_render_batch_strided(self._ctx, scene.handle, arr.ctypes.data_as(C.POINTER(_capi.SgsCamera)),
                                                           b, C.byref(cfg), r0, r1, ptr, frame_stride, stats, self._stream()), self._ctx)
        if want_stats:
            return ret, [s.as_dict() for s in stats]
        return ret

    def pack_rgba8(self, rgb: torch.Tensor, tonemap: Optional[str] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float32 [H,W,3] -> uint8 [H,W,4] (alpha 255): the array shape cam.get_rgba() returns.  `out`: a contiguous uint8
        [H,W,4] tensor on the device to write into (e.g. a band of a frame buffer).

        tonemap="reinhard" applies x / (1 + x) first — the operator the reference's stage selects
        (Data/template.usda:102,196).  An optional host-side op, off by default and outside the
        parity contract: the simulator's exposure/white-point settings are not reproduced, frames stay linear otherwise."""
        if tonemap is not None:
            if tonemap != "reinhard":
                raise ValueError("tonemap must be None or 'reinhard'")
            rgb = rgb / (1.0 + rgb.clamp_min(0.0))
        h, w = int(rgb.shape[0]), int(rgb.shape[1])
        if out is None:
            out = torch.empty((h, w, 4), dtype=torch.uint8, device=self.device)
        elif (out.device != self.device or out.dtype != torch.uint8 or not out.is_contiguous() or tuple(out.shape) != (h, w, 4)):
            raise ValueError("out must be a contiguous uint8 [H,W,4] tensor on the renderer's device")
        self._lib.check(self._lib.sgs_pack_rgba8(self._ctx, rgb.contiguous().data_ptr(), out.data_ptr(), w, h,
                                                 self._stream()), self._ctx)
        return out

    # -- host frames (the boundary the reference really has: uint8 arrays on the host) -----------------------------------------
    def host_frames(self, shape, depth: int = 2) -> "HostFrames":
        """A ring of `depth` PINNED uint8 host buffers of `shape` (+ their device-side uint8 twins, a copy stream and events):
        what get_rgba()-shaped callers read frames from.  Cached per (shape, depth)."""
        key = (tuple(int(v) for v in shape), int(depth))
        ring = self._host_rings.get(key) if hasattr(self, "_host_rings") else None
        if ring is None:
            if not hasattr(self, "_host_rings"):
                self._host_rings = {}
            ring = self._host_rings[key] = HostFrames(self, key[0], key[1])
        return ring

    def render_rgba8_host(self, camera: Camera, gaussians, *, config: Optional[RenderConfig] = None, tonemap: Optional[str] = None) -> np.ndarray:
        """One frame as the reference's callers receive it: uint8 [H,W,4] (alpha 255) in HOST memory (simple_env.py:1380-1386;
        generate_images.py:428-431).  Render and pack are enqueued back to back on the current stream — the pack kernel writes the
        pinned host buffer itself (no device twin, no copy engine, no pageable staging copy) — and the call waits once.  The array is renderer-owned
        and stays valid until the next-but-one call at this resolution (the reference's callers copy what they keep)."""
        scene = self._scene_of(gaussians)
        ring = self.host_frames((camera.height, camera.width, 4))
        rgb = self.render(camera, scene, config=config, out=ring.rgb_scratch(), sync=False)
        h = ring.submit(rgb, tonemap=tonemap, direct=True)
        try:
            self.sync()                  # the frame's status — and, stream-ordered behind it, pack + copy
        except _capi.SgsError as e:
            if e.code != -4:             # SGS_ERR_OVERFLOW: an asynchronous frame cannot grow the queues; a synchronous one does
                raise
            h.wait()
            rgb = self.render(camera, scene, config=config, out=ring.rgb_scratch(), sync=True)
            h = ring.submit(rgb, tonemap=tonemap, direct=True)
        return h.wait()

    # -- test hooks -------------------------------------------------------------------------------
    def debug_buffer(self, what, dtype):
        have = self._lib.sgs_debug_read(self._ctx, what, None, 0)
        if have < 0:
            self._lib.check(int(have), self._ctx)
        buf = np.zeros(int(have) // np.dtype(dtype).itemsize, dtype)
        if have:
            self._lib.sgs_debug_read(self._ctx, what, buf.ctypes.data, have)
        return buf

    def intermediates(self):
        """(tile_offsets, per-tile sorted Gaussian ids, Gaussian id of each slot, splat words [N_v,12])."""
        off = self.debug_buffer(_capi.BUF_TILE_OFFSETS, np.uint32).astype(np.int64)
        slots = self.debug_buffer(_capi.BUF_SORTED_SLOTS, np.uint32)
        ids = self.debug_buffer(_capi.BUF_SLOT_IDS, np.uint32)
        splats = self.debug_buffer(_capi.BUF_SPLATS, np.uint32).reshape(-1, 12)
        live = ids != 0xFFFFFFFF
        return off, ids[slots].astype(np.int64), ids[live].astype(np.int64), splats[live]

    def set_record_capacity(self, n: int):
        self._lib.check(self._lib.sgs_set_record_capacity(self._ctx, int(n)), self._ctx)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.sgs_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostFrames:
    """`depth` pinned uint8 host buffers + device twins of one shape, a D2H copy stream and one event per buffer.
    submit(rgb) packs fp32 -> uint8 RGBA on the CURRENT stream and starts the copy into the next pinned buffer on the copy
    stream (behind an event), so that the copy of frame i overlaps whatever the current stream does next (the rendering of
    frame i+1); wait() on the returned handle blocks until THAT copy has landed and returns the numpy view."""

    class Handle:
        def __init__(self, ring, k, n):
            self._ring, self._k, self._n = ring, k, n

        def wait(self) -> np.ndarray:
            self._ring._done[self._k].synchronize()
            a = self._ring._host_np[self._k]
            return a if self._n is None else a[:self._n]

    def __init__(self, renderer: "Renderer", shape, depth: int):
        self._r, self.shape, self.depth = renderer, tuple(shape), int(depth)
        dev = renderer.device
        self._dev = [torch.empty(self.shape, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        self._host = [torch.empty(self.shape, dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
        self._host_np = [t.numpy() for t in self._host]
        self._packed = [torch.cuda.Event() for _ in range(self.depth)]
        self._done = [torch.cuda.Event() for _ in range(self.depth)]
        self._copy_stream = torch.cuda.Stream(device=dev)
        self._turn = 0
        self._rgb = None

    def rgb_scratch(self) -> torch.Tensor:
        """An fp32 [..., 3] device buffer of the ring's image shape to render into (one: the pack consumes it in stream order)."""
        if self._rgb is None:
            self._rgb = torch.zeros(self.shape[:-1] + (3,), dtype=torch.float32, device=self._r.device)
        return self._rgb

    def submit(self, rgb: torch.Tensor, tonemap: Optional[str] = None, n: Optional[int] = None, direct: bool = False) -> "HostFrames.Handle":
        """rgb: fp32 [..., 3] of the ring's image shape (a batch [B,H,W,3] counts as one tall image).  n: leading entries that are
        valid (a partial last batch).
        direct=True: the pack kernel writes the PINNED buffer itself, over the link, on the current stream — no device twin, no copy
        engine, no second stream: for ONE frame somebody is waiting for (get_rgba(): 640x480 0.366 -> 0.339 ms, 1024x768 0.418 -> 0.387,
        profiles/r06u); a sweep's batches keep the copy stream, whose transfer runs beside the next batch's kernels."""
        k = self._turn
        self._turn = (k + 1) % self.depth
        self._done[k].synchronize()                  # the buffer's previous copy (depth frames ago) has long landed
        flat = rgb.reshape(-1, rgb.shape[-2], 3)
        if direct:
            if tonemap is not None:
                if tonemap != "reinhard":
                    raise ValueError("tonemap must be None or 'reinhard'")
                flat = flat / (1.0 + flat.clamp_min(0.0))
            r = self._r
            r._lib.check(r._lib.sgs_pack_rgba8(r._ctx, flat.contiguous().data_ptr(), self._host[k].data_ptr(), int(flat.shape[1]), int(flat.shape[0]),
                                               r._stream()), r._ctx)
            self._done[k].record(torch.cuda.current_stream(r.device))
            return HostFrames.Handle(self, k, n)
        self._r.pack_rgba8(flat, tonemap=tonemap, out=self._dev[k].reshape(-1, rgb.shape[-2], 4))
        cur = torch.cuda.current_stream(self._r.device)
        self._packed[k].record(cur)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(self._packed[k])
            self._host[k].copy_(self._dev[k], non_blocking=True)
            self._done[k].record(self._copy_stream)
        return HostFrames.Handle(self, k, n)


_default = {}


def default_renderer(device=None) -> Renderer:
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _default:
        _default[key] = Renderer(torch.device("cuda", key))
    return _default[key]


def render(camera: Camera, gaussians, *, config: Optional[RenderConfig] = None,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The drop-in call surface named by BASELINE.json: one frame, float32 [H,W,3] on the scene's GPU."""
    if isinstance(gaussians, Scene):             # an uploaded scene is drawn by the renderer (GPU) that holds it
        return gaussians._r.render(camera, gaussians, config=config, out=out)
    dev = gaussians.means.device if isinstance(gaussians, Gaussians) else None
    if dev is not None and dev.type != "cuda":
        dev = None
    return default_renderer(dev).render(camera, gaussians, config=config, out=out)
