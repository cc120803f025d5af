"""3DGS scene files -> renderer inputs.

SAGE-3D scenes reach the renderer as `3dgs_compressed.ply` -> (splat-transform) -> standard 3DGS `.ply`
-> (3dgrut ply_to_usd) -> USDZ (README.md:197-253); neither tool is vendored in the reference.  This module
reads the two PLY flavours directly so that real InteriorGS scenes can replace the synthetic ones:

  * `load_ply`   — the standard INRIA-layout binary PLY (x,y,z, f_dc_0..2, f_rest_*, opacity, scale_0..2,
                   rot_0..3; any property order, any SH degree 0..3) with the usual activations applied:
                   exp(scale), normalised quaternion (rot_0 = w), sigmoid(opacity), f_rest channel-major
                   -> sh[N, K, 3].
  * `save_ply`   — the inverse (for fixtures and round-trip tests).
  * `load_compressed_ply` / `save_compressed_ply` — the PlayCanvas "compressed.ply" layout (per-256-splat
                   chunk bounds + 11/10/11-bit position and log-scale, 2+10+10+10-bit "smallest three"
                   rotation, 8-bit colour/opacity, optional 8-bit SH).  Written from the published format
                   description and checked against a file assembled by hand from that layout
                   (tests/test_next_rows.py); files produced by splat-transform itself could not be tried — the
                   tool is not available offline.
Pure NumPy, host side; the arrays go to `Renderer.upload` like any other scene.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Optional, Tuple

import numpy as np

SH_C0 = 0.28209479177387814
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def _read_header(f):
    magic = f.readline().strip()
    if magic != b"ply":
        raise ValueError("not a PLY file")
    fmt, elements, cur = None, [], None
    while True:
        line = f.readline()
        if not line:
            raise ValueError("unterminated PLY header")
        tok = line.decode("ascii", "replace").split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            cur = {"name": tok[1], "count": int(tok[2]), "props": []}
            elements.append(cur)
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError("list properties are not supported")
            cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
    if fmt not in ("binary_little_endian", "binary_big_endian"):
        raise ValueError(f"unsupported PLY format {fmt!r} (binary only)")
    return fmt, elements


def read_elements(path) -> Dict[str, np.ndarray]:
    """All elements of a binary PLY as structured arrays."""
    out = {}
    with open(path, "rb") as f:
        fmt, elements = _read_header(f)
        end = "<" if fmt == "binary_little_endian" else ">"
        for el in elements:
            dt = np.dtype([(n, end + t) for n, t in el["props"]])
            buf = f.read(dt.itemsize * el["count"])
            if len(buf) != dt.itemsize * el["count"]:
                raise ValueError(f"truncated PLY: element {el['name']}")
            out[el["name"]] = np.frombuffer(buf, dtype=dt, count=el["count"])
    return out


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def load_ply(path):
    """Standard 3DGS PLY -> (means, scales, quats(wxyz), opacities, sh[N,K,3], sh_degree), float32, activated."""
    v = read_elements(path)["vertex"]
    names = v.dtype.names
    n = v.shape[0]
    col = lambda k: v[k].astype(np.float32)
    means = np.stack([col("x"), col("y"), col("z")], 1)
    scales = np.exp(np.stack([col("scale_0"), col("scale_1"), col("scale_2")], 1))
    quats = np.stack([col("rot_0"), col("rot_1"), col("rot_2"), col("rot_3")], 1)
    quats /= np.maximum(np.linalg.norm(quats, axis=1, keepdims=True), 1e-20)
    opac = _sigmoid(col("opacity"))
    rest = sorted((k for k in names if re.fullmatch(r"f_rest_\d+", k)), key=lambda k: int(k.split("_")[-1]))
    k_rest = len(rest) // 3
    deg = {0: 0, 3: 1, 8: 2, 15: 3}.get(k_rest)
    if deg is None or len(rest) != 3 * k_rest:
        raise ValueError(f"unexpected number of f_rest properties: {len(rest)}")
    sh = np.empty((n, k_rest + 1, 3), np.float32)
    sh[:, 0, :] = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1)
    if k_rest:
        r = np.stack([col(k) for k in rest], 1).reshape(n, 3, k_rest)       # channel-major in the file
        sh[:, 1:, :] = np.transpose(r, (0, 2, 1))
    return means, scales.astype(np.float32), quats.astype(np.float32), opac.astype(np.float32), sh, deg


def save_ply(path, means, scales, quats, opacities, sh, sh_degree):
    """Inverse of load_ply: de-activates (log scale, logit opacity) and writes the INRIA property layout."""
    means, scales, quats, sh = (np.asarray(a, np.float32) for a in (means, scales, quats, sh))
    o = np.clip(np.asarray(opacities, np.float64), 1e-7, 1 - 1e-7)
    n, k = means.shape[0], (sh_degree + 1) ** 2
    props = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * (k - 1))] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    arr = np.zeros(n, dtype=[(p, "<f4") for p in props])
    arr["x"], arr["y"], arr["z"] = means.T
    for c in range(3):
        arr[f"f_dc_{c}"] = sh[:, 0, c]
    if k > 1:
        r = np.transpose(sh[:, 1:, :], (0, 2, 1)).reshape(n, 3 * (k - 1))
        for i in range(3 * (k - 1)):
            arr[f"f_rest_{i}"] = r[:, i]
    arr["opacity"] = np.log(o / (1 - o)).astype(np.float32)
    for c in range(3):
        arr[f"scale_{c}"] = np.log(scales[:, c])
    for c in range(4):
        arr[f"rot_{c}"] = quats[:, c]
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
        for p in props:
            f.write(f"property float {p}\n".encode())
        f.write(b"end_header\n")
        f.write(arr.tobytes())


# ---- PlayCanvas compressed.ply (experimental, see module docstring) -----------------------------------
_SQRT2 = math.sqrt(2.0)


def _unpack_unorm(v, bits):
    return (v & ((1 << bits) - 1)).astype(np.float32) / float((1 << bits) - 1)


def _unpack_111011(p):
    return np.stack([_unpack_unorm(p >> 21, 11), _unpack_unorm(p >> 11, 10), _unpack_unorm(p, 11)], 1)


def _pack_111011(u):
    q = np.round(np.clip(u, 0, 1) * np.array([2047, 1023, 2047])).astype(np.uint32)
    return (q[:, 0] << 21) | (q[:, 1] << 11) | q[:, 2]


def load_compressed_ply(path, sh_decode: Optional[str] = None):
    """sh_decode: decode_sh_bytes' mode — REQUIRED when the file has an `sh` element (no default, see decode_sh_bytes)."""
    el = read_elements(path)
    if "sh" in el and sh_decode is None:
        raise ValueError('this compressed.ply carries 8-bit SH coefficients: pass sh_decode="bin_centre", "linear255" or "bin_centre_ends" '
                         "(what the tool that wrote / would decompress the file uses; there is no default)")
    ch, v = el["chunk"], el["vertex"]
    n = v.shape[0]
    ci = np.arange(n) // 256
    lerp = lambda u, lo, hi: lo + u * (hi - lo)
    cmin = lambda p: np.stack([ch[f"min_{p}{a}"] for a in ("x", "y", "z")], 1).astype(np.float32)[ci]
    cmax = lambda p: np.stack([ch[f"max_{p}{a}"] for a in ("x", "y", "z")], 1).astype(np.float32)[ci]
    means = lerp(_unpack_111011(v["packed_position"]), cmin(""), cmax(""))
    scales = np.exp(lerp(_unpack_111011(v["packed_scale"]), cmin("scale_"), cmax("scale_")))
    pr = v["packed_rotation"]
    a = (_unpack_unorm(pr >> 20, 10) - 0.5) * _SQRT2
    b = (_unpack_unorm(pr >> 10, 10) - 0.5) * _SQRT2
    c = (_unpack_unorm(pr, 10) - 0.5) * _SQRT2
    m = np.sqrt(np.maximum(0.0, 1.0 - (a * a + b * b + c * c)))
    which = (pr >> 30).astype(np.int64)
    quats = np.empty((n, 4), np.float32)                          # (w, x, y, z); `which` = index of the dropped (largest) one
    order = {0: (m, a, b, c), 1: (a, m, b, c), 2: (a, b, m, c), 3: (a, b, c, m)}
    for w_, comps in order.items():
        sel = which == w_
        for k in range(4):
            quats[sel, k] = comps[k][sel]
    pc = v["packed_color"]
    rgba = np.stack([_unpack_unorm(pc >> 24, 8), _unpack_unorm(pc >> 16, 8), _unpack_unorm(pc >> 8, 8), _unpack_unorm(pc, 8)], 1)
    if "min_r" in ch.dtype.names:
        lo = np.stack([ch["min_r"], ch["min_g"], ch["min_b"]], 1).astype(np.float32)[ci]
        hi = np.stack([ch["max_r"], ch["max_g"], ch["max_b"]], 1).astype(np.float32)[ci]
        rgb = lerp(rgba[:, :3], lo, hi)
    else:
        rgb = rgba[:, :3]
    dc = (rgb - 0.5) / SH_C0
    opac = rgba[:, 3]
    if "sh" in el:
        s = el["sh"]
        rest = sorted(s.dtype.names, key=lambda k: int(k.split("_")[-1]))
        k_rest = len(rest) // 3
        r = decode_sh_bytes(np.stack([s[k] for k in rest], 1), sh_decode)
        sh = np.empty((n, k_rest + 1, 3), np.float32)
        sh[:, 0] = dc
        sh[:, 1:] = np.transpose(r.reshape(n, 3, k_rest), (0, 2, 1))
        deg = {3: 1, 8: 2, 15: 3}[k_rest]
    else:
        sh, deg = dc[:, None, :].astype(np.float32), 0
    return means.astype(np.float32), scales.astype(np.float32), quats, opac.astype(np.float32), sh, deg


CHUNK_PROPS = ["min_x", "min_y", "min_z", "max_x", "max_y", "max_z",
               "min_scale_x", "min_scale_y", "min_scale_z", "max_scale_x", "max_scale_y", "max_scale_z"]
CHUNK_COLOR_PROPS = ["min_r", "min_g", "min_b", "max_r", "max_g", "max_b"]
PACKED_PROPS = ["packed_position", "packed_rotation", "packed_scale", "packed_color"]


def decode_sh_bytes(v, mode: str = "bin_centre"):
    """8-bit SH coefficient(s) -> float32, three readings (include/sage_gs.h SGS_SH_DECODE_*; the device applies the same arithmetic):
      "bin_centre"       (v / 256 - 0.5) * 8 + 4 / 256 = v / 32 - 4 + 1 / 64 — the centre of the bin trunc((x / 8 + 0.5) * 256) that
                         encode_compressed (and, as far as this repo can tell, the PlayCanvas writer) puts x into.  Exact in fp32.
      "linear255"        v * 8 / 255 - 4: the end codes are -4 and +4 (to this builder's recollection what the PlayCanvas readers apply).
      "bin_centre_ends"  bin centres, but 0 -> -4 and 255 -> +4 exactly.
    They differ by at most 1/64 per coefficient.  The reference delegates this decode to @playcanvas/splat-transform (README.md:197-231),
    which is neither vendored nor installable offline, so the tool's choice could not be pinned here — hence NO default anywhere a file's
    bytes are read (load_compressed_ply, Renderer.upload_compressed, sgs_compressed_scene.sh_decode): the caller says which."""
    v = np.asarray(v)
    if mode == "linear255":
        return (v.astype(np.float64) * (8.0 / 255.0) - 4.0).astype(np.float32)      # in double, as a JavaScript converter computes it; the device does the same
    r = (v.astype(np.float32) / 256.0 - 0.5) * 8.0 + (4.0 / 256.0)
    if mode == "bin_centre_ends":
        r = np.where(v == 0, np.float32(-4.0), np.where(v == 255, np.float32(4.0), r)).astype(np.float32)
    elif mode != "bin_centre":
        raise ValueError('mode must be "bin_centre", "linear255" or "bin_centre_ends"')
    return r


def read_compressed_payload(path):
    """A compressed.ply as the C ABI takes it (include/sage_gs.h sgs_compressed_scene): (chunks float32 [nch,18], packed uint32 [n,4],
    sh uint8 [n, 3 k_rest] or None, sh_degree) — no dequantisation here: `Renderer.upload_compressed` hands these to the device, whose
    layout kernel decodes them (csrc/sgs_kernels.h unpack_gaussian).  `load_compressed_ply` below is the same arithmetic in NumPy."""
    el = read_elements(path)
    ch, v = el["chunk"], el["vertex"]
    nch = ch.shape[0]
    chunks = np.empty((nch, 18), np.float32)
    for k, name in enumerate(CHUNK_PROPS):
        chunks[:, k] = ch[name]
    if "min_r" in ch.dtype.names:
        for k, name in enumerate(CHUNK_COLOR_PROPS):
            chunks[:, 12 + k] = ch[name]
    else:
        chunks[:, 12:15] = 0.0; chunks[:, 15:18] = 1.0
    packed = np.stack([v[name].astype(np.uint32) for name in PACKED_PROPS], 1)
    if "sh" in el:
        s = el["sh"]
        rest = sorted(s.dtype.names, key=lambda k: int(k.split("_")[-1]))
        sh = np.ascontiguousarray(np.stack([s[k] for k in rest], 1).astype(np.uint8))
        deg = {3: 1, 8: 2, 15: 3}[len(rest) // 3]
    else:
        sh, deg = None, 0
    return np.ascontiguousarray(chunks), np.ascontiguousarray(packed), sh, deg


def encode_compressed(means, scales, quats, opacities, sh, sh_degree):
    """(chunk table [nch,12] as a structured array, vertex table, sh bytes [n, 3 k_rest] or None): the quantisation of the PlayCanvas
    layout — per 256-Gaussian chunk the bounds of position and log-scale, 11/10/11-bit position and scale, 2+10+10+10-bit "smallest three"
    rotation, 8-bit colour (0.5 + C0 dc) and opacity, and — degree > 0 — 8-bit SH (value / 8 + 0.5, 256 levels, truncated)."""
    means, scales, quats = (np.asarray(a, np.float64) for a in (means, scales, quats))
    sh = np.asarray(sh, np.float64)
    n = means.shape[0]
    nch = (n + 255) // 256
    ci = np.arange(n) // 256
    ls = np.log(scales)
    rgb = np.clip(0.5 + SH_C0 * sh[:, 0, :], 0, 1)
    chunk = np.zeros(nch, dtype=[(k, "<f4") for k in CHUNK_PROPS])

    def bounds(a, pre):
        ap = np.concatenate([a, np.repeat(a[-1:], nch * 256 - n, 0)]).reshape(nch, 256, 3) if n else np.zeros((0, 256, 3))
        lo, hi = ap.min(1), ap.max(1)
        for k, ax in enumerate("xyz"):
            chunk[f"min_{pre}{ax}"] = lo[:, k]; chunk[f"max_{pre}{ax}"] = hi[:, k]
        span = np.where(hi > lo, hi - lo, 1.0)
        return (a - lo[ci]) / span[ci]
    up, us = bounds(means, ""), bounds(ls, "scale_")
    q = quats / np.linalg.norm(quats, axis=1, keepdims=True)
    which = np.argmax(np.abs(q), axis=1)
    q = q * np.sign(q[np.arange(n), which])[:, None]
    keep = np.ones((n, 4), bool); keep[np.arange(n), which] = False
    rest = q[keep].reshape(n, 3)                                               # the three that are kept, in (w,x,y,z) order
    u = np.round(np.clip(rest / _SQRT2 + 0.5, 0, 1) * 1023).astype(np.uint32)
    vert = np.zeros(n, dtype=[(k, "<u4") for k in PACKED_PROPS])
    vert["packed_position"] = _pack_111011(up)
    vert["packed_scale"] = _pack_111011(us)
    vert["packed_rotation"] = (which.astype(np.uint32) << 30) | (u[:, 0] << 20) | (u[:, 1] << 10) | u[:, 2]
    c8 = np.round(np.concatenate([rgb, np.clip(np.asarray(opacities, np.float64), 0, 1)[:, None]], 1) * 255).astype(np.uint32)
    vert["packed_color"] = (c8[:, 0] << 24) | (c8[:, 1] << 16) | (c8[:, 2] << 8) | c8[:, 3]
    shb = None
    k_rest = (sh_degree + 1) ** 2 - 1
    if k_rest > 0:
        r = np.transpose(sh[:, 1:, :], (0, 2, 1)).reshape(n, 3 * k_rest)          # channel-major, as f_rest_* are
        shb = np.clip(np.trunc((r / 8.0 + 0.5) * 256.0), 0, 255).astype(np.uint8)
    return chunk, vert, shb


def save_compressed_ply(path, means, scales, quats, opacities, sh, sh_degree):
    """Encoder matching load_compressed_ply / read_compressed_payload (fixtures, round-trip tests, bench.py's compressed upload)."""
    chunk, vert, shb = encode_compressed(means, scales, quats, opacities, sh, sh_degree)
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement chunk %d\n" % chunk.shape[0]).encode())
        for k in chunk.dtype.names:
            f.write(f"property float {k}\n".encode())
        f.write(("element vertex %d\n" % vert.shape[0]).encode())
        for k in vert.dtype.names:
            f.write(f"property uint {k}\n".encode())
        if shb is not None:
            f.write(("element sh %d\n" % shb.shape[0]).encode())
            for k in range(shb.shape[1]):
                f.write(f"property uchar f_rest_{k}\n".encode())
        f.write(b"end_header\n")
        f.write(chunk.tobytes()); f.write(vert.tobytes())
        if shb is not None:
            f.write(shb.tobytes())


def to_gaussians(arrays: Tuple, device, model_to_world=None):
    """(means, scales, quats, opacities, sh, degree) -> renderer.Gaussians on `device`."""
    import torch
    from .renderer import Gaussians
    m, s, q, o, sh, deg = arrays
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
    return Gaussians(t(m), t(s), t(q), t(o), t(sh), int(deg), model_to_world)
