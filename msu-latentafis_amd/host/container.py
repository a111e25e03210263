"""Packed gallery container `AFISGAL1` — Python mirror of csrc/template_io.{h,cpp}.

One mmap-able file with the staged gallery's SoA arrays (only rolled minutiae template 0 and texture template 0 of every file,
texture counts clamped to 1000, CSR offsets), every section 64-byte aligned; little-endian:
  0   char[8] "AFISGAL1" | 8  u32 version=1, des_len=96, code_len=16, 0 | 24  i64 G, n_minutiae, n_texture_points, names_bytes
  56  u64 offset[13]: minu_off i64[G+1] | tex_off i64[G+1] | empty u8[G] | minu_x i16[] | minu_y i16[] | minu_ori f32[] |
      minu_des f32[][96] | tex_x i16[] | tex_y i16[] | tex_ori f32[] | tex_codes u8[][16] | name_off i64[G+1] | names
The reference has no such file: it re-reads every rolled .dat for every (latent, rolled) pair (matching/matcher.cpp:173,:278)."""
import struct
from typing import List, Optional, Sequence

import numpy as np

from .synth import PackedGallery

MAGIC = b"AFISGAL1"
_DT = ["<i8", "<i8", "u1", "<i2", "<i2", "<f4", "<f4", "<i2", "<i2", "<f4", "u1", "<i8", "S1"]


def _sections(g: PackedGallery, names: Optional[Sequence[str]]):
    G = g.G
    blob = b"".join(n.encode() for n in names) if names else b""
    name_off = np.zeros(G + 1, np.int64)
    if names:
        name_off[1:] = np.cumsum([len(n.encode()) for n in names])
    empty = ((np.diff(g.minu_off) == 0) & (np.diff(g.tex_off) == 0)).astype(np.uint8)
    return [np.asarray(g.minu_off, "<i8"), np.asarray(g.tex_off, "<i8"), empty, np.asarray(g.minu_x, "<i2"), np.asarray(g.minu_y, "<i2"),
            np.asarray(g.minu_ori, "<f4"), np.asarray(g.minu_des, "<f4").reshape(-1), np.asarray(g.tex_x, "<i2"), np.asarray(g.tex_y, "<i2"),
            np.asarray(g.tex_ori, "<f4"), np.asarray(g.tex_codes, np.uint8).reshape(-1), name_off, np.frombuffer(blob, np.uint8)]


def write_container(path: str, g: PackedGallery, names: Optional[Sequence[str]] = None) -> None:
    if names is not None and len(names) != g.G:
        raise ValueError("names must be one per template")
    if (np.diff(g.tex_off) > 1000).any():
        raise ValueError("texture templates must be clamped to 1000 points (matcher.cpp:546-547)")
    secs = _sections(g, names)
    pos = (56 + 8 * 13 + 63) // 64 * 64
    offs = []
    for a in secs:
        offs.append(pos)
        pos = (pos + a.nbytes + 63) // 64 * 64
    hdr = MAGIC + struct.pack("<4I", 1, 96, 16, 0) + struct.pack("<4q", g.G, len(g.minu_x), len(g.tex_x), secs[12].nbytes) + struct.pack("<13Q", *offs)
    with open(path, "wb") as f:
        f.write(hdr)
        at = len(hdr)
        for o, a in zip(offs, secs):
            f.write(b"\0" * (o - at)); f.write(a.tobytes()); at = o + a.nbytes


def read_container(path: str, first: int = 0, count: int = -1):
    """-> (PackedGallery of templates [first, first+count), names, texture point count of EVERY template in the file)."""
    mm = np.memmap(path, np.uint8, "r")
    if mm[:8].tobytes() != MAGIC:
        raise ValueError(f"{path}: not an AFISGAL1 container")
    ver, dl, cl, _ = struct.unpack("<4I", mm[8:24].tobytes())
    G, NM, NT, NB = struct.unpack("<4q", mm[24:56].tobytes())
    offs = struct.unpack("<13Q", mm[56:56 + 104].tobytes())
    if ver != 1 or dl != 96 or cl != 16:
        raise ValueError(f"{path}: unsupported container geometry")
    if count < 0:
        count = G - first
    if first < 0 or first + count > G:
        raise ValueError("template range outside the container")
    sizes = [G + 1, G + 1, G, NM, NM, NM, NM * 96, NT, NT, NT, NT * 16, G + 1, NB]

    def sec(i):
        dt = np.dtype(_DT[i] if i != 12 else "u1")
        return np.frombuffer(mm, dt, sizes[i], offs[i])
    mo, to = sec(0), sec(1)
    m0, m1, t0, t1 = mo[first], mo[first + count], to[first], to[first + count]
    g = PackedGallery(minu_off=(mo[first:first + count + 1] - m0).astype(np.int64), minu_x=sec(3)[m0:m1].copy(), minu_y=sec(4)[m0:m1].copy(),
                      minu_ori=sec(5)[m0:m1].copy(), minu_des=sec(6)[m0 * 96:m1 * 96].reshape(-1, 96).copy(),
                      tex_off=(to[first:first + count + 1] - t0).astype(np.int64), tex_x=sec(7)[t0:t1].copy(), tex_y=sec(8)[t0:t1].copy(),
                      tex_ori=sec(9)[t0:t1].copy(), tex_codes=sec(10)[t0 * 16:t1 * 16].reshape(-1, 16).copy())
    no, blob = sec(11), sec(12).tobytes()
    names = [blob[no[i]:no[i + 1]].decode() for i in range(first, first + count)]
    return g, names, np.diff(to).astype(np.int32)
