#!/usr/bin/env python3
"""Pack the reference's OBJ meshes into compact binary fixtures under assets/.

The GPU box has no /root/reference, and the benchmark / parity scenes need the meshes the
reference ships (assets/teapot.obj, assets/bunny.obj, assets/pear/obj/pear_export.obj).  Those
files are *data* (Utah teapot, Stanford bunny, a pear scan), not source; they are stored here as
binary32 vertex tables + index tables ("RTXM1", see raytracer_amd/assets.py for the reader), which
is exactly the information the reference's loader extracts from the text
(/root/reference/src/gfx/mesh.c:107-160: `v`, `vt`, `vn`, `f` records only).  Numbers are parsed
with libc strtof — what sscanf("%f") does — so every float is bit-identical to the reference's
parse; raytracer_amd/assets.py re-emits OBJ text with %.9g, which strtof maps back to the same
bits.

Run once in the build container:  python tools/pack_assets.py
"""
import ctypes
import os
import struct
import sys

import numpy as np

REF = os.environ.get("RTX_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")

libc = ctypes.CDLL("libc.so.6")
libc.strtof.restype = ctypes.c_float
libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]


def strtof(tok: str) -> float:
    return libc.strtof(tok.encode(), None)


def pack(src: str, dst: str) -> None:
    v, vt, vn, counts, corners = [], [], [], [], []
    kinds = 0  # bit0: vt present, bit1: vn present
    with open(src, "r") as fd:
        for line in fd:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v" and len(tok) >= 4:
                v.append([strtof(t) for t in tok[1:4]])
            elif tok[0] == "vt" and len(tok) >= 3:
                vt.append([strtof(t) for t in tok[1:3]])
            elif tok[0] == "vn" and len(tok) >= 4:
                vn.append([strtof(t) for t in tok[1:4]])
            elif tok[0] == "f":
                cs = []
                for t in tok[1:]:
                    parts = t.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    if ti:
                        kinds |= 1
                    if ni:
                        kinds |= 2
                    cs.append((vi, ti, ni))
                assert len(cs) in (3, 4), f"{src}: face with {len(cs)} corners"
                counts.append(len(cs))
                corners.extend(cs)
    v = np.asarray(v, np.float32).reshape(-1, 3)
    vt = np.asarray(vt, np.float32).reshape(-1, 2)
    vn = np.asarray(vn, np.float32).reshape(-1, 3)
    counts = np.asarray(counts, np.uint8)
    corners = np.asarray(corners, np.int32).reshape(-1, 3)
    cols = [0] + ([1] if kinds & 1 else []) + ([2] if kinds & 2 else [])
    corners = np.ascontiguousarray(corners[:, cols])
    with open(dst, "wb") as fd:
        fd.write(struct.pack("<5sBHIIIII", b"RTXM1", kinds, 0, len(v), len(vt), len(vn), len(counts), len(corners)))
        for arr in (v, vt, vn, counts, corners):
            fd.write(arr.tobytes())
    tris = int((counts == 3).sum() + 2 * (counts == 4).sum())
    print(f"{os.path.basename(dst)}: {len(v)} v, {len(vt)} vt, {len(vn)} vn, {len(counts)} faces -> {tris} tris, "
          f"{os.path.getsize(dst)} bytes")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    pack(f"{REF}/assets/teapot.obj", f"{OUT}/teapot.mesh")
    pack(f"{REF}/assets/bunny.obj", f"{OUT}/bunny.mesh")
    pack(f"{REF}/assets/pear/obj/pear_export.obj", f"{OUT}/pear.mesh")
    sys.exit(0)
