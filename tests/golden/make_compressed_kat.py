#!/usr/bin/env python3
"""Known-answer vectors for the PlayCanvas "compressed.ply" fields (the format InteriorGS ships, README.md:197-231 of the
reference): packed words written out BY HAND from the published bit layout, expected values computed here in exact rational arithmetic
(fractions) + correctly rounded square roots / exponentials (mpmath-free: decimal with 50 digits) — NOT with sage_gs.ply or the kernels.
    python tests/golden/make_compressed_kat.py > tests/golden/compressed_ply_kat.json

Layout (little-endian uint32 per field):
  packed_position / packed_scale   bits 31..21 = x (11 bits), 20..11 = y (10 bits), 10..0 = z (11 bits); value = field / (2^bits - 1),
                                   then min + value * (max - min) with the chunk's bounds (positions; LOG scales -> exp)
  packed_rotation                  bits 31..30 = index of the component that was dropped (the largest, made positive); 29..20, 19..10, 9..0 =
                                   the other three in (w,x,y,z) order, (field / 1023 - 0.5) * sqrt(2); dropped = sqrt(1 - sum of squares)
  packed_color                     bits 31..24 r, 23..16 g, 15..8 b, 7..0 opacity, / 255; f_dc = (c - 0.5) / 0.28209479177387814 (with the
                                   chunk's colour range, if any: c = min + value (max - min))
  sh bytes                         three candidate readings, all listed per byte (the writer stores trunc((f / 8 + 0.5) * 256)):
                                   bin centre (byte / 256 - 0.5) * 8 + 4 / 256, linear byte * 8 / 255 - 4, bin centre with exact ends;
                                   f_rest_k is channel-major: k = channel * k_rest + (coefficient - 1)

Nothing in this file imports sage_gs: the expected values are NOT produced by the repo's encoder or decoder (ply.encode_compressed,
ply.load_compressed_ply, the layout kernel) — those are what the vectors are held against.
"""
import json
from decimal import Decimal, getcontext
from fractions import Fraction as F

getcontext().prec = 50
C0 = Decimal("0.28209479177387814")


def dec(fr):
    return Decimal(fr.numerator) / Decimal(fr.denominator)


def field(word, hi, bits):
    return (word >> (hi - bits + 1)) & ((1 << bits) - 1)


def unorm3(word):
    return [F(field(word, 31, 11), 2047), F(field(word, 20, 10), 1023), F(field(word, 10, 11), 2047)]


cases = []
chunk = {"min": [F(-3), F(1, 2), F(-10)], "max": [F(5), F(9, 2), F(-2)],
         "min_scale": [F(-6), F(-5), F(-7)], "max_scale": [F(-1), F(0), F(-2)],
         "min_rgb": [F(0), F(0), F(0)], "max_rgb": [F(1), F(1), F(1)]}
for word in (0x00000000, 0xFFFFFFFF, 0x80100400, 0x12345678, 0xFFE00000, 0x001FF800, 0x000007FF, 0xABCDEF01):
    u = unorm3(word)
    pos = [chunk["min"][k] + u[k] * (chunk["max"][k] - chunk["min"][k]) for k in range(3)]
    ls = [chunk["min_scale"][k] + u[k] * (chunk["max_scale"][k] - chunk["min_scale"][k]) for k in range(3)]
    cases.append({"field": "packed_position", "word": f"0x{word:08X}", "expect": [float(dec(v)) for v in pos]})
    cases.append({"field": "packed_scale", "word": f"0x{word:08X}", "expect": [float(dec(v).exp()) for v in ls]})
for word in (0x1FF7FDFF, 0x00000000, 0x7FFFFFFF, 0x9AB12CDE, 0xC0080200, 0xFFFFFFFF, 0x5A5A5A5A):
    which = word >> 30
    r2 = Decimal(2).sqrt()
    abc = [(Decimal(field(word, 29, 10)) / 1023 - Decimal("0.5")) * r2, (Decimal(field(word, 19, 10)) / 1023 - Decimal("0.5")) * r2,
           (Decimal(field(word, 9, 10)) / 1023 - Decimal("0.5")) * r2]
    rest = Decimal(1) - sum(v * v for v in abc)
    big = rest.sqrt() if rest > 0 else Decimal(0)
    q = abc[:which] + [big] + abc[which:]
    cases.append({"field": "packed_rotation", "word": f"0x{word:08X}", "expect": [float(v) for v in q]})          # (w, x, y, z)
for word in (0x00000000, 0xFFFFFFFF, 0x80808080, 0x10C0FF7F, 0xFF000001, 0x336699CC):
    c = [Decimal(field(word, 31, 8)) / 255, Decimal(field(word, 23, 8)) / 255, Decimal(field(word, 15, 8)) / 255]
    cases.append({"field": "packed_color", "word": f"0x{word:08X}", "expect_dc": [float((v - Decimal("0.5")) / C0) for v in c],
                  "expect_opacity": float(Decimal(field(word, 7, 8)) / 255)})
# An SH byte has THREE candidate readings (include/sage_gs.h SGS_SH_DECODE_*; which one @playcanvas/splat-transform applies cannot be pinned
# offline, so the caller must say): the centre of the writer's truncation bin, the linear map onto [-4, 4], and bin centres with exact ends.
# All three in exact arithmetic, end codes included:
for byte in (0, 1, 127, 128, 200, 254, 255):
    centre = (Decimal(byte) / 256 - Decimal("0.5")) * 8 + Decimal(4) / 256
    cases.append({"field": "sh_byte", "byte": byte, "expect": float(centre),
                  "expect_linear255": float(Decimal(byte) * 8 / 255 - 4),
                  "expect_bin_centre_ends": float(Decimal(-4) if byte == 0 else Decimal(4) if byte == 255 else centre)})
# ... and WHERE a byte goes: the file's f_rest_k properties are channel-major (k = channel * 15 + coefficient - 1 at degree 3); the renderer
# takes [coefficient][channel].  One vertex whose 45 bytes are 5 k + 3: coefficient c (1..15), channel ch must decode byte 5 (15 ch + c - 1) + 3.
cases.append({"field": "sh_order", "bytes": [5 * k + 3 for k in range(45)],
              "expect": [[float((Decimal(5 * (15 * ch + c) + 3) / 256 - Decimal("0.5")) * 8 + Decimal(4) / 256) for ch in range(3)] for c in range(15)]})
print(json.dumps({"chunk": {k: [float(dec(x)) for x in v] for k, v in chunk.items()}, "cases": cases,
                  "note": "tests/golden/make_compressed_kat.py: hand-written words, exact arithmetic; fp32 decoders must agree to ~1e-6 relative"}, indent=1))
