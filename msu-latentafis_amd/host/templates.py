"""Template / codebook data model and the reference's on-disk formats (host side, numpy only).

Formats follow the only writers whose layout matches the reference C++ reader:
  * latent  .dat : extraction/descriptor_PQ.py:80-175  (Template2Bin_Byte_latent)
  * rolled  .dat : extraction/descriptor_PQ.py:178-272 (Template2Bin_Byte_PQ_rolled)
  * codebook.dat : matching/matcher.cpp:74-93          (3 x int16 header M,K,dsub + M*K*dsub float32)
and are read back the way matching/matcher.cpp:785-983 reads them (little endian):

  12 x u16 header ([0] = version 1) | u16 h, w, blkH, blkW | u8 nMinuTpl |
  per minutiae template { u16 n; if n > 0: u16 x[n], u16 y[n], f32 ori[n], u16 des_len, f32 des[n*des_len] } |
  u8 nTexTpl |
  per texture template  { u16 n; if n > 0: u16 x[n], u16 y[n], f32 ori[n], u16 des_len,
                          latent: f32 des[n*des_len]   /   rolled: u8 codes[n*des_len] }

Texture x, y are BLOCK coordinates ((pixel - 24) / 16, descriptor_PQ.py:152,156,250,254); minutiae x, y are
pixels.  This module is the Python mirror of msu-latentafis_amd/csrc/template_io.{h,cpp}; the C++ one is what
the `match` CLI and the C-ABI use.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

MAX_NROF_MINUTIAE = 2000   # matcher.cpp:788 / descriptor_PQ.py:84
MAX_DES_LENGTH = 192       # matcher.cpp:789
BAD_DES_LENGTH = 8         # parser return code without a reference counterpart
DESCRIPTOR_NORM = 1.73     # extraction/descriptor_DR.py:150-152 rescales every descriptor to this L2 norm


@dataclass
class Codebook:
    """PQ codebook: M sub-quantizers x K codewords x dsub dims (matcher.cpp:31-94)."""
    words: np.ndarray  # float32 [M, K, dsub]

    @property
    def M(self) -> int: return self.words.shape[0]
    @property
    def K(self) -> int: return self.words.shape[1]
    @property
    def dsub(self) -> int: return self.words.shape[2]

    @staticmethod
    def from_bytes(buf: bytes) -> "Codebook":
        M, K, dsub = struct.unpack_from("<3h", buf, 0)
        n = M * K * dsub
        if n <= 0 or len(buf) < 6 + 4 * n:
            raise ValueError("codebook is empty!")
        w = np.frombuffer(buf, dtype="<f4", count=n, offset=6).reshape(M, K, dsub).copy()
        return Codebook(w)

    @staticmethod
    def load(path: str) -> "Codebook":
        with open(path, "rb") as f:
            return Codebook.from_bytes(f.read())

    def to_bytes(self) -> bytes:
        return struct.pack("<3h", self.M, self.K, self.dsub) + self.words.astype("<f4").tobytes()

    def encode(self, des: np.ndarray) -> np.ndarray:
        """Nearest codeword per sub-space (squared L2, first minimum), as TrainedPQEncoder.encode_multi
        (descriptor_PQ.py:19-27, scipy.cluster.vq.vq)."""
        des = np.asarray(des, dtype=np.float32)
        n = des.shape[0]
        codes = np.empty((n, self.M), dtype=np.uint8)
        for m in range(self.M):
            sub = des[:, m * self.dsub:(m + 1) * self.dsub].astype(np.float64)
            d = ((sub[:, None, :] - self.words[m][None].astype(np.float64)) ** 2).sum(-1)
            codes[:, m] = d.argmin(1)
        return codes

    def encode_fast(self, des: np.ndarray) -> np.ndarray:
        """Nearest codeword per sub-space through |c|^2 - 2 x.c in float32 (one GEMM per sub-space): for GENERATORS of synthetic templates, where a different choice between
        two all but equidistant codewords does not matter; `encode` is the reference's routine."""
        des = np.asarray(des, dtype=np.float32)
        codes = np.empty((des.shape[0], self.M), dtype=np.uint8)
        for m in range(self.M):
            w = self.words[m]
            c2 = np.einsum("kd,kd->k", w, w)
            for a in range(0, des.shape[0], 1 << 16):
                x = des[a:a + (1 << 16), m * self.dsub:(m + 1) * self.dsub]
                codes[a:a + (1 << 16), m] = (c2[None, :] - 2.0 * (x @ w.T)).argmin(1)
        return codes

    @staticmethod
    def synthetic(seed: int = 0, M: int = 16, K: int = 256, dsub: int = 6) -> "Codebook":
        """Random codebook with the shipped file's statistics (values ~N(0, 0.164))."""
        rng = np.random.default_rng(seed)
        return Codebook((rng.standard_normal((M, K, dsub)) * 0.164).astype(np.float32))


@dataclass
class MinutiaeTemplate:
    x: np.ndarray      # [n] pixel coords (stored u16, read as i16)
    y: np.ndarray
    ori: np.ndarray    # float32 [n]
    des: np.ndarray    # float32 [n, des_len]

    @property
    def n(self) -> int: return int(len(self.x))


@dataclass
class TextureTemplate:
    x: np.ndarray      # [n] BLOCK coords
    y: np.ndarray
    ori: np.ndarray
    des: Optional[np.ndarray] = None     # latent: float32 [n, 96]
    codes: Optional[np.ndarray] = None   # rolled: uint8 [n, 16]

    @property
    def n(self) -> int: return int(len(self.x))


@dataclass
class FPTemplate:
    minu: List[MinutiaeTemplate] = field(default_factory=list)
    tex: List[TextureTemplate] = field(default_factory=list)
    h: int = 800
    w: int = 768
    blkH: int = 50
    blkW: int = 48


def _header(t: FPTemplate, version: int = 1) -> bytes:
    hdr = [0] * 12
    hdr[0] = version
    return struct.pack("<12H", *hdr) + struct.pack("<4H", t.h, t.w, min(t.blkH, 50), min(t.blkW, 50))


def _points(x, y, ori) -> bytes:
    return (np.asarray(x).astype("<u2").tobytes() + np.asarray(y).astype("<u2").tobytes()
            + np.asarray(ori).astype("<f4").tobytes())


def _minu_block(t: FPTemplate) -> bytes:
    out = [struct.pack("<B", len(t.minu))]
    for m in t.minu:
        n = min(m.n, MAX_NROF_MINUTIAE)
        out.append(struct.pack("<H", n))
        if n <= 0:
            continue
        out.append(_points(m.x[:n], m.y[:n], m.ori[:n]))
        des = np.asarray(m.des[:n], dtype="<f4")
        out.append(struct.pack("<H", des.shape[1]))
        out.append(des.tobytes())
    return b"".join(out)


def write_latent(t: FPTemplate) -> bytes:
    """Template2Bin_Byte_latent (descriptor_PQ.py:80-175)."""
    if len(t.minu) == 0:
        return struct.pack("<12H", 1, *([0] * 11)) + struct.pack("<4H", 0, 0, 0, 0)
    out = [_header(t), _minu_block(t), struct.pack("<B", len(t.tex))]
    for x in t.tex:
        n = min(x.n, MAX_NROF_MINUTIAE)
        out.append(struct.pack("<H", n))
        if n > 0:
            out.append(_points(x.x[:n], x.y[:n], x.ori[:n]))
            des = np.asarray(x.des[:n], dtype="<f4")
            out.append(struct.pack("<H", des.shape[1]))
            out.append(des.tobytes())
    return b"".join(out)


def write_rolled(t: FPTemplate) -> bytes:
    """Template2Bin_Byte_PQ_rolled (descriptor_PQ.py:178-272)."""
    if len(t.minu) == 0:
        return struct.pack("<12H", 1, *([0] * 11)) + struct.pack("<4H", 0, 0, 0, 0)
    out = [_header(t), _minu_block(t), struct.pack("<B", len(t.tex))]
    for x in t.tex:
        n = min(x.n, MAX_NROF_MINUTIAE)
        out.append(struct.pack("<H", n))
        if n > 0:
            out.append(_points(x.x[:n], x.y[:n], x.ori[:n]))
            codes = np.asarray(x.codes[:n], dtype=np.uint8)
            out.append(struct.pack("<H", codes.shape[1]))
            out.append(codes.tobytes())
    return b"".join(out)


class _Reader:
    """ifstream-like: a read past the end delivers what is left, later reads deliver nothing."""

    def __init__(self, buf: bytes):
        self.buf, self.pos, self.fail = buf, 0, False

    def take(self, n: int) -> bytes:
        if self.fail:
            return b""
        b = self.buf[self.pos:self.pos + n]
        self.pos += len(b)
        if len(b) < n:
            self.fail = True
        return b

    def scalar(self, fmt: str, default=0):
        b = self.take(struct.calcsize(fmt))
        if len(b) < struct.calcsize(fmt):
            return default
        return struct.unpack(fmt, b)[0]

    def array(self, dtype: str, n: int) -> np.ndarray:
        item = np.dtype(dtype).itemsize
        b = self.take(item * n)
        a = np.zeros(n, dtype=dtype)
        k = len(b) // item
        a[:k] = np.frombuffer(b[:k * item], dtype=dtype)
        return a


def _read(buf: bytes, rolled: bool) -> (int, FPTemplate):
    """Mirror of Matcher::load_FP_template (matcher.cpp:785-884 latent, :886-983 rolled).
    Returns (rc, template); rc as the reference: 0 ok, 1 empty, 2 too many minutiae, -1 texture too large; 8 (no reference
    counterpart) = a descriptor length outside 1..192, parsing stopped there."""
    t = FPTemplate()
    if (rolled and len(buf) <= 10) or (not rolled and len(buf) <= 0):
        return 1, t
    r = _Reader(buf)
    r.take(24)
    t.h = r.scalar("<h"); t.w = r.scalar("<h"); t.blkH = min(r.scalar("<h"), 50); t.blkW = min(r.scalar("<h"), 50)
    nmt = r.scalar("<B")
    for _ in range(nmt):
        n = r.scalar("<h")
        if r.fail:
            break
        if n <= 0:
            continue                      # zero-minutiae templates are dropped: later indices shift
        if n > MAX_NROF_MINUTIAE:
            return 2, t
        x = r.array("<i2", n); y = r.array("<i2", n); ori = r.array("<f4", n)
        dl = r.scalar("<h")
        if r.fail:
            break
        if dl <= 0 or dl > MAX_DES_LENGTH:
            return BAD_DES_LENGTH, t      # no reference counterpart (it overruns a stack buffer): stop, keep what was parsed
        des = r.array("<f4", n * dl).reshape(n, dl)
        t.minu.append(MinutiaeTemplate(x, y, ori, des))
    ntt = r.scalar("<B")
    for _ in range(ntt):
        n = r.scalar("<h")
        if r.fail:
            break
        if n <= 0:
            continue
        if n > MAX_NROF_MINUTIAE:
            return -1, t
        x = r.array("<i2", n); y = r.array("<i2", n); ori = r.array("<f4", n)
        dl = r.scalar("<h")
        if r.fail:
            break
        if dl <= 0 or dl > MAX_DES_LENGTH:
            return BAD_DES_LENGTH, t
        if rolled:
            # the reference reads n*dl floats (4x over-read to EOF, matcher.cpp:975) and keeps n*dl bytes
            raw = r.array("u1", n * dl * 4)
            t.tex.append(TextureTemplate(x, y, ori, codes=raw[:n * dl].reshape(n, dl).copy()))
        else:
            t.tex.append(TextureTemplate(x, y, ori, des=r.array("<f4", n * dl).reshape(n, dl)))
    return 0, t


def read_latent(buf: bytes):
    return _read(buf, rolled=False)


def read_rolled(buf: bytes):
    rc, t = _read(buf, rolled=True)
    if rc < 0:                            # matcher.cpp:173-177
        t.minu, t.tex = [], []
    return rc, t
