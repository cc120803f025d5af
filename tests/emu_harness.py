"""Builds and drives the product's kernels under the wave64 emulator (tests/hipemu) — CPU tests only.

The emulated library is the product's own csrc/sgs_api.hip + csrc/sgs_kernels.h compiled by g++
against tests/hipemu/hip/hip_runtime.h.  It exists so that `-m "not gpu"` tests can check the kernel
LOGIC (compaction, duplication, radix sort, composite) against the oracle; it is never the thing
measured or shipped, and the product package cannot load it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sage-3d_official_amd")
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libsage_gs_emu.so")
SRCS = [os.path.join(PKG, "csrc", n) for n in ("sgs_api.hip", "sgs_kernels.h", "sgs_common.h")] + \
       [os.path.join(EMU_DIR, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "sage_gs.h")]

if PKG not in sys.path:
    sys.path.insert(0, PKG)
from sage_gs import _capi  # noqa: E402


def build_emu(force=False):
    os.makedirs(os.path.dirname(EMU_LIB), exist_ok=True)
    stale = force or not os.path.exists(EMU_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in SRCS)
    if stale:
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O2", "-g", "-fopenmp", "-fPIC", "-shared",
                               "-I", EMU_DIR, SRCS[0], "-o", EMU_LIB])
    return EMU_LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _capi.Lib(build_emu())
    return _lib


class EmuRenderer:
    """Minimal numpy-facing driver of the C ABI (host pointers: the emulator's 'device' is the host)."""

    def __init__(self, record_capacity=1 << 20):
        self.lib = lib()
        self.ctx = C.c_void_p()
        self.lib.check(self.lib.sgs_create(0, _capi.BACKEND_HIP, C.byref(self.ctx)))
        self.lib.check(self.lib.sgs_set_record_capacity(self.ctx, record_capacity), self.ctx)
        self.scene = None

    def upload(self, means, scales, quats, opacities, sh, sh_degree):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (means, scales, quats, opacities, sh)]
        if self.scene is not None:
            self.lib.sgs_scene_free(self.ctx, self.scene)
        sc = C.c_void_p()
        self.lib.check(self.lib.sgs_scene_upload(self.ctx, arrs[0].shape[0], sh_degree,
                                                 *[a.ctypes.data for a in arrs], 0, C.byref(sc)), self.ctx)
        self.scene = sc
        self.n = arrs[0].shape[0]

    def upload_compressed(self, chunks, packed, sh, sh_degree, sh_decode=None):
        """sgs_scene_upload_compressed with host buffers (ply.read_compressed_payload's arrays)."""
        c = np.ascontiguousarray(chunks, np.float32); p = np.ascontiguousarray(packed, np.uint32)
        b = None if sh is None else np.ascontiguousarray(sh, np.uint8)
        if self.scene is not None:
            self.lib.sgs_scene_free(self.ctx, self.scene)
        z = _capi.SgsCompressedScene(p.shape[0], c.shape[0], int(sh_degree), 0 if sh_decode is None else sh_decode if isinstance(sh_decode, int) else _capi.SH_DECODE[sh_decode], c.ctypes.data, p.ctypes.data, b.ctypes.data if b is not None else None)
        sc = C.c_void_p()
        self.lib.check(self.lib.sgs_scene_upload_compressed(self.ctx, C.byref(z), 0, C.byref(sc)), self.ctx)
        self.scene, self.n = sc, p.shape[0]

    def scene_geom(self):
        """float [N,11] of the scene as the device holds it (after one frame): mean, opacity, scale, quaternion wxyz."""
        return self.debug(_capi.BUF_SCENE_GEOM, np.float32).reshape(-1, 11)

    def scene_sh(self):
        """float [N, K, 3] of the scene's SH coefficients as the projection kernel evaluates them (after one frame)."""
        return self.debug(_capi.BUF_SCENE_SH, np.float32).reshape(self.n, -1, 3)

    def render(self, cam, cfg=None, rows=(0, -1), out=None, flags=0, full_sort=False, loose_cull=False, interleave=None,
               chunk_cull=True, stats=True, deep=True, fine=True):
        flags |= 0 if chunk_cull else _capi.FLAG_NO_CHUNK_CULL
        flags |= 0 if fine else _capi.FLAG_NO_FINE_TILES
        flags |= 0 if deep else _capi.FLAG_NO_DEEP
        flags |= _capi.FLAG_STATS if stats else 0
        flags |= _capi.FLAG_FULL_SORT if full_sort else 0
        flags |= _capi.FLAG_LOOSE_CULL if loose_cull else 0
        c = _capi.make_camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                              np.asarray(cam.view, np.float32).reshape(4, 4).tolist())
        k = self.lib.default_config()
        if cfg is not None:
            k.near_z, k.far_z, k.dilation, k.clamp = cfg.near, cfg.far, cfg.dilation, cfg.clamp
            k.alpha_min, k.alpha_max, k.t_min = cfg.alpha_min, cfg.alpha_max, cfg.t_min
            for i in range(3):
                k.bg[i] = cfg.background[i]
            k.sh_degree = cfg.sh_degree
        k.flags = flags
        if interleave is not None:         # (stride, phase): compact image of the owned tile rows
            k.tile_row_stride, k.tile_row_phase = interleave
            owned = len(range(interleave[1], (cam.height + 15) // 16, interleave[0]))
            out = np.full((16 * owned, cam.width, 3), -1.0, np.float32)
        if out is None:
            out = np.zeros((cam.height, cam.width, 3), np.float32)
        st = _capi.SgsStats()
        self.lib.check(self.lib.sgs_render(self.ctx, self.scene, C.byref(c), C.byref(k), rows[0], rows[1],
                                           out.ctypes.data, C.byref(st), None), self.ctx)
        return out, st.as_dict()

    def render_batch(self, cams, fine=True):
        """sgs_render_batch: the frames of `cams` (one resolution) as frame groups; [n, H, W, 3]."""
        arr = (_capi.SgsCamera * len(cams))(*[_capi.make_camera(c.width, c.height, c.fx, c.fy, c.cx, c.cy,
                                                                np.asarray(c.view, np.float32).reshape(4, 4).tolist()) for c in cams])
        k = self.lib.default_config()
        k.flags = 0 if fine else _capi.FLAG_NO_FINE_TILES
        out = np.zeros((len(cams), cams[0].height, cams[0].width, 3), np.float32)
        self.lib.check(self.lib.sgs_render_batch(self.ctx, self.scene, arr, len(cams), C.byref(k), 0, -1, out.ctypes.data, None, None), self.ctx)
        return out

    def render_aux(self, cam, cfg=None, fine=True):
        c = _capi.make_camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                              np.asarray(cam.view, np.float32).reshape(4, 4).tolist())
        k = self.lib.default_config()
        k.flags = 0 if fine else _capi.FLAG_NO_FINE_TILES
        out = np.zeros((cam.height, cam.width, 3), np.float32)
        aux = np.zeros((cam.height, cam.width, 2), np.float32)
        st = _capi.SgsStats()
        self.lib.check(self.lib.sgs_render_rgbd(self.ctx, self.scene, C.byref(c), C.byref(k), 0, -1, out.ctypes.data,
                                                aux.ctypes.data, C.byref(st), None), self.ctx)
        return out, aux

    def set_record_capacity(self, n):
        self.lib.check(self.lib.sgs_set_record_capacity(self.ctx, int(n)), self.ctx)

    def set_tuning(self, **kw):
        t = _capi.SgsTuning()
        self.lib.check(self.lib.sgs_get_tuning(self.ctx, C.byref(t)), self.ctx)
        for k, v in kw.items():
            setattr(t, k, float(v) if k == "fine_tile_growth" else int(v))
        self.lib.check(self.lib.sgs_set_tuning(self.ctx, C.byref(t)), self.ctx)

    def tuning(self):
        t = _capi.SgsTuning()
        self.lib.check(self.lib.sgs_get_tuning(self.ctx, C.byref(t)), self.ctx)
        return {k: (float if k == "fine_tile_growth" else int)(getattr(t, k)) for k, _ in t._fields_}

    def chunk_skipped(self):
        return self.debug(_capi.BUF_CHUNK_SKIPPED, np.uint8)

    def row_records(self, n_rows, reset=True):
        out = np.zeros(int(n_rows), np.int64)
        self.lib.check(self.lib.sgs_row_records(self.ctx, out.ctypes.data, int(n_rows), 1 if reset else 0), self.ctx)
        return out

    def debug(self, what, dtype, count_hint=None):
        have = self.lib.sgs_debug_read(self.ctx, what, None, 0)
        if have < 0:
            self.lib.check(int(have), self.ctx)
        buf = np.zeros(int(have) // np.dtype(dtype).itemsize, dtype)
        self.lib.sgs_debug_read(self.ctx, what, buf.ctypes.data, have)
        return buf

    def intermediates(self):
        """(tile_offsets, per-tile sorted Gaussian ids, splat table keyed by Gaussian id)."""
        off = self.debug(_capi.BUF_TILE_OFFSETS, np.uint32)
        slots = self.debug(_capi.BUF_SORTED_SLOTS, np.uint32)
        ids = self.debug(_capi.BUF_SLOT_IDS, np.uint32)
        splats = self.debug(_capi.BUF_SPLATS, np.uint32).reshape(-1, 12)
        live = ids != 0xFFFFFFFF
        return off.astype(np.int64), ids[slots].astype(np.int64), ids[live].astype(np.int64), splats[live]

    def close(self):
        if self.scene is not None:
            self.lib.sgs_scene_free(self.ctx, self.scene)
            self.scene = None
        if self.ctx:
            self.lib.sgs_destroy(self.ctx)
            self.ctx = C.c_void_p()
