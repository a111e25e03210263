"""ctypes binding of libafis_hip.so (include/afis_matcher.h) with the reference's `PQ::Matcher` surface.

Mirrors matching/matcher.h:35-51: Matcher(code_file), One2List_matching, List2List_matching, plus the in-memory
calls the C ABI adds (gallery add/commit, batched search).  There is NO CPU fallback: if the HIP library is missing
or no gfx950 device is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .templates import Codebook, FPTemplate, read_latent, read_rolled

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libafis_hip.so")            # the product: include/afis_matcher.h, nothing else
TEST_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libafis_hip_test.so")  # the same objects + the parity taps (include/afis_matcher_taps.h); tests only


class AfisError(RuntimeError):
    pass


class MinutiaeView(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.POINTER(C.c_int16)), ("y", C.POINTER(C.c_int16)), ("ori", C.POINTER(C.c_float)),
                ("des_len", C.c_int32), ("des", C.POINTER(C.c_float))]


class TextureView(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.POINTER(C.c_int16)), ("y", C.POINTER(C.c_int16)), ("ori", C.POINTER(C.c_float)),
                ("des_len", C.c_int32), ("des", C.POINTER(C.c_float)), ("codes", C.POINTER(C.c_uint8))]


class TemplateView(C.Structure):
    _fields_ = [("n_minu", C.c_int32), ("minu", C.POINTER(MinutiaeView)), ("n_tex", C.c_int32), ("tex", C.POINTER(TextureView))]


class Timing(C.Structure):
    _fields_ = [("lut_ms", C.c_float), ("adc_ms", C.c_float), ("tex_tail_ms", C.c_float), ("minu_ms", C.c_float), ("fuse_ms", C.c_float), ("topk_ms", C.c_float),
                ("total_ms", C.c_float), ("adc_launches", C.c_int32), ("adc_lookups", C.c_int64), ("pairs", C.c_int64), ("adc_bound_ms", C.c_float), ("adc_refine_ms", C.c_float),
                ("cands_ms", C.c_float), ("minu_graph_ms", C.c_float), ("launch_groups", C.c_int32), ("overlapped_groups", C.c_int32),
                ("minu_tasks", C.c_int64), ("minu_fallback_tasks", C.c_int64), ("minu_tasks_small", C.c_int64), ("minu_tasks_medium", C.c_int64), ("minu_tasks_large", C.c_int64),
                ("bound_clock_ghz", C.c_float), ("cands_clock_ghz", C.c_float)]


EXPORTS = ["afis_create", "afis_create_from_codebook", "afis_device_info", "afis_destroy", "afis_last_error", "afis_gallery_add", "afis_gallery_add_dat", "afis_gallery_add_dat_batch", "afis_gallery_reserve",
           "afis_gallery_add_packed", "afis_gallery_commit", "afis_gallery_size", "afis_gallery_save", "afis_gallery_load",
           "afis_gallery_file_info", "afis_gallery_file_names", "afis_rank_list", "afis_search", "afis_search_dat", "afis_queries_upload",
           "afis_search_resident", "afis_queries_free", "afis_correspondences", "afis_match_all_templates", "afis_pq_encode", "afis_encode_rolled_dat", "afis_get_timing", "afis_get_timing2", "afis_set_option", "afis_get_option"]
# include/afis_matcher_taps.h: exported by libafis_hip_test.so only
TAP_EXPORTS = ["afis_debug_lut", "afis_debug_texture_rowmax", "afis_debug_stage_list", "afis_debug_phase_cycles", "afis_debug_atan2_grid", "afis_debug_graph_arith", "afis_debug_refine_stats"]


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise AfisError(f"{path} not found: build it with `make -C msu-latentafis_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32p, i64p, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)
    lib.afis_create.argtypes = [C.POINTER(vp), fp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.afis_create_from_codebook.argtypes = [C.POINTER(vp), C.c_char_p, C.c_size_t, C.c_int]
    lib.afis_destroy.argtypes = [vp]; lib.afis_destroy.restype = None
    lib.afis_last_error.argtypes = [vp]; lib.afis_last_error.restype = C.c_char_p
    lib.afis_rank_list.argtypes = [C.POINTER(C.c_float), C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    lib.afis_gallery_add.argtypes = [vp, C.POINTER(TemplateView), C.c_int]
    lib.afis_gallery_add_dat.argtypes = [vp, C.c_char_p, C.c_size_t, i32p]
    lib.afis_gallery_reserve.argtypes = [vp, C.c_int64]
    if hasattr(lib, "afis_gallery_add_dat_batch"):
        lib.afis_gallery_add_dat_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int64, i32p]
    lib.afis_gallery_add_packed.argtypes = [vp, C.c_int64, i64p, C.POINTER(C.c_int16), C.POINTER(C.c_int16), fp, fp,
                                            i64p, C.POINTER(C.c_int16), C.POINTER(C.c_int16), fp, C.POINTER(C.c_uint8)]
    lib.afis_gallery_commit.argtypes = [vp, C.c_int64]
    lib.afis_gallery_save.argtypes = [vp, C.c_char_p, C.POINTER(C.c_char_p)]
    lib.afis_gallery_load.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64]
    lib.afis_gallery_file_info.argtypes = [C.c_char_p, i64p, i64p, i64p, i32p]
    lib.afis_gallery_file_names.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.afis_gallery_size.argtypes = [vp]; lib.afis_gallery_size.restype = C.c_int64
    lib.afis_search.argtypes = [vp, C.POINTER(TemplateView), C.c_int, fp, fp, i32p, C.c_int, i64p, fp]
    lib.afis_search_dat.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, fp, fp, i32p, C.c_int, i64p, fp]
    lib.afis_queries_upload.argtypes = [vp, C.POINTER(TemplateView), C.c_int, C.POINTER(vp)]
    lib.afis_search_resident.argtypes = [vp, vp, fp, fp, i32p, C.c_int, i64p, fp]
    lib.afis_correspondences.argtypes = [vp, vp, i64p, C.c_int, i32p, C.POINTER(C.c_int16)]
    lib.afis_queries_free.argtypes = [vp, vp]; lib.afis_queries_free.restype = None
    lib.afis_match_all_templates.argtypes = [vp, vp, fp, i32p, i32p]
    lib.afis_pq_encode.argtypes = [vp, fp, C.c_int64, C.POINTER(C.c_uint8)]
    lib.afis_encode_rolled_dat.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), i32p]
    lib.afis_get_timing.argtypes = [vp, C.POINTER(Timing)]
    if hasattr(lib, "afis_get_timing2"):                                # absent from older builds compared by tools/lib_ab.py
        lib.afis_get_timing2.argtypes = [vp, C.POINTER(Timing), C.c_size_t]
    lib.afis_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    if hasattr(lib, "afis_device_info"):                                # round 5
        lib.afis_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, i32p]
    if hasattr(lib, "afis_get_option"):
        lib.afis_get_option.argtypes = [vp, C.c_char_p, i64p]
    if hasattr(lib, "afis_debug_stage_list"):                           # the parity taps: libafis_hip_test.so (and older builds) only
        lib.afis_debug_stage_list.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, fp, i32p, i32p, i32p]
        lib.afis_debug_lut.argtypes = [vp, C.POINTER(TemplateView), fp, i32p]
        lib.afis_debug_texture_rowmax.argtypes = [vp, C.POINTER(TemplateView), C.c_int64, fp, i32p, i32p]
        lib.afis_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        lib.afis_debug_atan2_grid.argtypes = [vp, C.c_int, fp]
    if hasattr(lib, "afis_debug_graph_arith"):
        lib.afis_debug_graph_arith.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    if hasattr(lib, "afis_debug_refine_stats"):
        lib.afis_debug_refine_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_int]
    return lib


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


class _Views:
    """Builds afis_template_view arrays from FPTemplate objects and keeps every backing array alive."""

    def __init__(self, templates: Sequence[FPTemplate]):
        self.keep = []
        self.arr = (TemplateView * max(1, len(templates)))()
        for i, t in enumerate(templates):
            mv = (MinutiaeView * max(1, len(t.minu)))()
            for j, m in enumerate(t.minu):
                x = np.ascontiguousarray(m.x, np.int16); y = np.ascontiguousarray(m.y, np.int16)
                o = np.ascontiguousarray(m.ori, np.float32); d = np.ascontiguousarray(m.des, np.float32)
                self.keep += [x, y, o, d]
                mv[j] = MinutiaeView(len(x), _ptr(x, C.c_int16), _ptr(y, C.c_int16), _ptr(o, C.c_float), d.shape[1] if d.ndim == 2 else 0, _ptr(d, C.c_float))
            tv = (TextureView * max(1, len(t.tex)))()
            for j, s in enumerate(t.tex):
                x = np.ascontiguousarray(s.x, np.int16); y = np.ascontiguousarray(s.y, np.int16); o = np.ascontiguousarray(s.ori, np.float32)
                self.keep += [x, y, o]
                des_p, codes_p, dl = None, None, 0
                if s.des is not None:
                    d = np.ascontiguousarray(s.des, np.float32); self.keep.append(d); des_p = _ptr(d, C.c_float); dl = d.shape[1]
                if s.codes is not None:
                    c = np.ascontiguousarray(s.codes, np.uint8); self.keep.append(c); codes_p = _ptr(c, C.c_uint8); dl = c.shape[1]
                tv[j] = TextureView(len(x), _ptr(x, C.c_int16), _ptr(y, C.c_int16), _ptr(o, C.c_float), dl, des_p, codes_p)
            self.keep += [mv, tv]
            self.arr[i] = TemplateView(len(t.minu), mv, len(t.tex), tv)
        self.n = len(templates)


class Matcher:
    """`PQ::Matcher` (matching/matcher.h:35) on one MI355X.  One instance = one device = one gallery shard."""

    def __init__(self, code_file, device: int = 0, lib_path: Optional[str] = None, taps: bool = False):
        """taps=True loads libafis_hip_test.so (the product objects + the afis_debug_* parity taps); the default is the product library."""
        self.lib = load_library(lib_path or (TEST_LIB_PATH if taps else LIB_PATH))
        self.has_taps = hasattr(self.lib, "afis_debug_stage_list")
        if isinstance(code_file, Codebook):
            buf = code_file.to_bytes()
        elif isinstance(code_file, (bytes, bytearray)):
            buf = bytes(code_file)
        else:
            with open(code_file, "rb") as f:
                buf = f.read()
        self.ctx = C.c_void_p()
        rc = self.lib.afis_create_from_codebook(C.byref(self.ctx), buf, len(buf), device)
        if rc != 0:
            self.ctx = None
            raise AfisError(f"afis_create failed ({rc}): {self.lib.afis_last_error(None).decode()}")
        self.gallery_files: List[str] = []

    def device_info(self, device: int) -> dict:
        """What the HIP runtime says about a device (afis_device_info): name, PCI bus id, UUID, compute units."""
        name = C.create_string_buffer(256); pci = C.create_string_buffer(64); uuid = C.create_string_buffer(40); ncu = C.c_int32(0)
        rc = self.lib.afis_device_info(device, name, 256, pci, 64, uuid, 40, C.byref(ncu))
        if rc != 0:
            raise AfisError(f"afis_device_info({device}) failed ({rc})")
        return {"name": name.value.decode(errors="replace"), "pci_bus_id": pci.value.decode(errors="replace"), "uuid": uuid.value.decode(errors="replace"), "compute_units": int(ncu.value)}

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.afis_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise AfisError(f"afis error {rc}: {self.lib.afis_last_error(self.ctx).decode()}")

    # ---- gallery ----------------------------------------------------------------------------------------------
    def gallery_add(self, templates: Sequence[FPTemplate]):
        v = _Views(templates)
        self._chk(self.lib.afis_gallery_add(self.ctx, v.arr, v.n))

    def gallery_add_dat(self, buf: bytes) -> int:
        rc = C.c_int32(0)
        self._chk(self.lib.afis_gallery_add_dat(self.ctx, buf, len(buf), C.byref(rc)))
        return rc.value

    def gallery_add_dat_batch(self, bufs: Sequence[bytes]) -> np.ndarray:
        """n rolled .dat files at once (parsed on the host's threads); returns the reader's code per file."""
        n = len(bufs)
        arr = (C.c_char_p * max(1, n))(*bufs)
        lens = (C.c_size_t * max(1, n))(*[len(b) for b in bufs])
        rc = np.zeros(max(1, n), np.int32)
        self._chk(self.lib.afis_gallery_add_dat_batch(self.ctx, arr, lens, n, _ptr(rc, C.c_int32)))
        return rc[:n]

    def gallery_add_packed(self, g):
        mo = np.ascontiguousarray(g.minu_off, np.int64); to = np.ascontiguousarray(g.tex_off, np.int64)
        a = [np.ascontiguousarray(g.minu_x, np.int16), np.ascontiguousarray(g.minu_y, np.int16), np.ascontiguousarray(g.minu_ori, np.float32),
             np.ascontiguousarray(g.minu_des, np.float32), np.ascontiguousarray(g.tex_x, np.int16), np.ascontiguousarray(g.tex_y, np.int16),
             np.ascontiguousarray(g.tex_ori, np.float32), np.ascontiguousarray(g.tex_codes, np.uint8)]
        self._chk(self.lib.afis_gallery_add_packed(self.ctx, len(mo) - 1, _ptr(mo, C.c_int64), _ptr(a[0], C.c_int16), _ptr(a[1], C.c_int16),
                                                   _ptr(a[2], C.c_float), _ptr(a[3], C.c_float), _ptr(to, C.c_int64), _ptr(a[4], C.c_int16),
                                                   _ptr(a[5], C.c_int16), _ptr(a[6], C.c_float), _ptr(a[7], C.c_uint8)))

    def gallery_save(self, path: str, names: Sequence[str] = None):
        """Write the staged gallery as one packed container (before gallery_commit)."""
        arr = None
        if names is not None:
            arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        self._chk(self.lib.afis_gallery_save(self.ctx, path.encode(), arr))

    def gallery_load(self, path: str, first: int = 0, count: int = -1):
        """Append templates [first, first+count) of a packed container to the staged gallery."""
        self._chk(self.lib.afis_gallery_load(self.ctx, path.encode(), first, count))

    def gallery_file_info(self, path: str):
        """-> (G, minutiae, texture points, per-template texture point counts) of a packed container."""
        G = C.c_int64(0); nm = C.c_int64(0); nt = C.c_int64(0)
        if self.lib.afis_gallery_file_info(path.encode(), C.byref(G), C.byref(nm), C.byref(nt), None) != 0:
            raise AfisError(self.lib.afis_last_error(None).decode())
        tc = np.zeros(max(1, G.value), np.int32)
        if self.lib.afis_gallery_file_info(path.encode(), None, None, None, _ptr(tc, C.c_int32)) != 0:
            raise AfisError(self.lib.afis_last_error(None).decode())
        return G.value, nm.value, nt.value, tc[:G.value]

    def gallery_file_names(self, path: str, first: int = 0, count: int = -1) -> List[str]:
        need = C.c_size_t(0)
        if self.lib.afis_gallery_file_names(path.encode(), first, count, None, 0, C.byref(need)) != 0:
            raise AfisError(self.lib.afis_last_error(None).decode())
        buf = C.create_string_buffer(max(1, need.value))
        if self.lib.afis_gallery_file_names(path.encode(), first, count, buf, need.value, C.byref(need)) != 0:
            raise AfisError(self.lib.afis_last_error(None).decode())
        return [b.decode() for b in buf.raw[:need.value].split(b"\0")[:-1]] if need.value else []

    def gallery_reserve(self, n_templates: int):
        """Hint: the staged gallery will grow to about n_templates templates (host arrays reserve room once)."""
        self._chk(self.lib.afis_gallery_reserve(self.ctx, n_templates))

    def gallery_commit(self, index_base: int = 0):
        self._chk(self.lib.afis_gallery_commit(self.ctx, index_base))

    @property
    def gallery_size(self) -> int:
        return int(self.lib.afis_gallery_size(self.ctx))

    def set_option(self, name: str, value: int):
        self._chk(self.lib.afis_set_option(self.ctx, name.encode(), value))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        if not hasattr(self.lib, "afis_get_option"):
            return 0
        self._chk(self.lib.afis_get_option(self.ctx, name.encode(), C.byref(v)))
        return int(v.value)

    # ---- search -----------------------------------------------------------------------------------------------
    def _alloc(self, nq, k, want_scores, want_parts):
        G = self.gallery_size
        scores = np.empty((nq, G), np.float32) if want_scores else None
        parts = np.empty((nq, G, 4), np.float32) if want_parts else None
        status = np.zeros(nq, np.int32)
        ti = np.empty((nq, k), np.int64) if k > 0 else None
        ts = np.empty((nq, k), np.float32) if k > 0 else None
        args = (_ptr(scores, C.c_float) if want_scores else None, _ptr(parts, C.c_float) if want_parts else None, _ptr(status, C.c_int32), k,
                _ptr(ti, C.c_int64) if k > 0 else None, _ptr(ts, C.c_float) if k > 0 else None)
        return scores, parts, status, ti, ts, args

    def search(self, latents: Sequence[FPTemplate], k: int = 24, want_scores: bool = True, want_parts: bool = False):
        v = _Views(latents)
        scores, parts, status, ti, ts, args = self._alloc(v.n, k, want_scores, want_parts)
        self._chk(self.lib.afis_search(self.ctx, v.arr, v.n, *args))
        return {"scores": scores, "parts": parts, "status": status, "topk_idx": ti, "topk_score": ts}

    def search_dat(self, bufs: Sequence[bytes], k: int = 24, want_scores: bool = True, want_parts: bool = False):
        n = len(bufs)
        arr = (C.c_char_p * max(1, n))(*bufs)
        lens = (C.c_size_t * max(1, n))(*[len(b) for b in bufs])
        scores, parts, status, ti, ts, args = self._alloc(n, k, want_scores, want_parts)
        self._chk(self.lib.afis_search_dat(self.ctx, arr, lens, n, *args))
        return {"scores": scores, "parts": parts, "status": status, "topk_idx": ti, "topk_score": ts}

    def rank_list(self, scores, k: int = 24, ref_order: bool = False):
        """afis_rank_list (matcher.cpp:306-309): the k best of a score column, score descending; ref_order: equal scores in the order the reference binary's std::sort leaves them
        (else by ascending index, as the search's own top-k)."""
        s = np.ascontiguousarray(scores, np.float32).reshape(-1)
        idx = np.full(max(k, 1), -1, np.int64); sc = np.zeros(max(k, 1), np.float32)
        self._chk(self.lib.afis_rank_list(_ptr(s, C.c_float) if len(s) else None, C.c_int64(len(s)), C.c_int(int(ref_order)), C.c_int(k), _ptr(idx, C.c_int64), _ptr(sc, C.c_float)))
        return idx[:k], sc[:k]

    def upload_queries(self, latents: Sequence[FPTemplate]):
        v = _Views(latents)
        h = C.c_void_p()
        self._chk(self.lib.afis_queries_upload(self.ctx, v.arr, v.n, C.byref(h)))
        return (h, v.n)

    def search_resident(self, handle, k: int = 24, want_scores: bool = False, want_parts: bool = False):
        h, n = handle
        scores, parts, status, ti, ts, args = self._alloc(n, k, want_scores, want_parts)
        self._chk(self.lib.afis_search_resident(self.ctx, h, *args))
        return {"scores": scores, "parts": parts, "status": status, "topk_idx": ti, "topk_score": ts}

    def correspondences(self, latent: FPTemplate, gallery_idx: Sequence[int]):
        """Surviving minutiae correspondences (matcher.cpp:497-505) of one latent against each listed gallery template:
        a list (one entry per gallery index) of three int16 arrays [n_s][4] = (lx, ly, rx, ry), one per selected template
        (None where the reference does not run that scorer)."""
        v = _Views([latent])
        n = len(gallery_idx)
        gi = np.asarray(gallery_idx, np.int64).reshape(-1)
        counts = np.zeros((max(n, 1), 3), np.int32); xy = np.zeros((max(n, 1), 3, 120, 4), np.int16)
        self._chk(self.lib.afis_correspondences(self.ctx, v.arr, _ptr(gi, C.c_int64) if n else None, n, _ptr(counts, C.c_int32), _ptr(xy, C.c_int16)))
        return [[xy[i, s, :counts[i, s]].copy() if counts[i, s] >= 0 else None for s in range(3)] for i in range(n)]

    def One2One_matching_all_templates(self, latent: FPTemplate):
        """matcher.cpp:339-374 against every gallery template: (query status, rolled status [G], scores [G][n_minu + n_tex])."""
        v = _Views([latent])
        G = self.gallery_size; width = len(latent.minu) + len(latent.tex)
        scores = np.zeros((max(G, 1), max(width, 1)), np.float32); rs = np.zeros(max(G, 1), np.int32); qs = C.c_int32(0)
        self._chk(self.lib.afis_match_all_templates(self.ctx, v.arr, _ptr(scores, C.c_float), _ptr(rs, C.c_int32), C.byref(qs)))
        return qs.value, rs[:G], scores[:G, :width] if width else np.zeros((G, 0), np.float32)

    def pq_encode(self, des: np.ndarray) -> np.ndarray:
        """TrainedPQEncoder.encode_multi (descriptor_PQ.py:19-27) on the device: [n][96] fp32 -> [n][16] u8."""
        des = np.ascontiguousarray(des, np.float32).reshape(-1, 96)
        codes = np.zeros((des.shape[0], 16), np.uint8)
        self._chk(self.lib.afis_pq_encode(self.ctx, _ptr(des, C.c_float), des.shape[0], _ptr(codes, C.c_uint8)))
        return codes

    def encode_rolled_dat(self, buf: bytes):
        """A template with fp32 texture descriptors (latent layout) -> (reader rc, the rolled-layout file with PQ codes)."""
        need = C.c_size_t(0); rc = C.c_int32(0)
        self._chk(self.lib.afis_encode_rolled_dat(self.ctx, buf, len(buf), None, 0, C.byref(need), C.byref(rc)))
        out = C.create_string_buffer(max(1, need.value))
        self._chk(self.lib.afis_encode_rolled_dat(self.ctx, buf, len(buf), out, need.value, C.byref(need), C.byref(rc)))
        return rc.value, out.raw[:need.value]

    def free_queries(self, handle):
        self.lib.afis_queries_free(self.ctx, handle[0])

    def timing(self) -> dict:
        t = Timing()
        if hasattr(self.lib, "afis_get_timing2"):
            self._chk(self.lib.afis_get_timing2(self.ctx, C.byref(t), C.sizeof(Timing)))
        else:
            self._chk(self.lib.afis_get_timing(self.ctx, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timing._fields_ if n != "reserved_"}

    def _tap(self, name: str):
        if not hasattr(self.lib, name):
            raise AfisError(f"{name} is a parity tap: construct Matcher(..., taps=True) (libafis_hip_test.so); the product library does not export it")
        return getattr(self.lib, name)

    def phase_cycles(self, reset: bool = True):
        out = (C.c_uint64 * 32)()
        self._chk(self._tap("afis_debug_phase_cycles")(self.ctx, out, 1 if reset else 0))
        return list(out)

    # ---- parity taps ------------------------------------------------------------------------------------------
    def debug_atan2_grid(self, R: int) -> np.ndarray:
        out = np.empty((2 * R + 1, 2 * R + 1), np.float32)
        self._chk(self._tap("afis_debug_atan2_grid")(self.ctx, R, _ptr(out, C.c_float)))
        return out

    def debug_graph_arith(self) -> list:
        out = (C.c_ulonglong * 8)()
        self._chk(self._tap("afis_debug_graph_arith")(self.ctx, out))
        return list(out)

    def refine_stats(self, reset: bool = True) -> dict:
        """adc_variant 9 with set_option("mf_stats", 1): what the selection / recomputation kernel did since the last reset."""
        out = (C.c_ulonglong * 8)()
        self._chk(self._tap("afis_debug_refine_stats")(self.ctx, out, 1 if reset else 0))
        return dict(zip(("pairs", "rows", "rows_evaluated", "cells_evaluated", "rows_evaluated_in_full", "bound_violations"), list(out)[:6]))

    def debug_stage_list(self, latent: FPTemplate, g: int, which: int, stage: int):
        """(sim, li, ri) of the scorer's correspondence list after a stage (None when the scorer is not run)."""
        v = _Views([latent])
        sim = np.zeros(200, np.float32); li = np.zeros(200, np.int32); ri = np.zeros(200, np.int32); n = C.c_int32(0)
        self._chk(self._tap("afis_debug_stage_list")(self.ctx, v.arr, g, which, stage, _ptr(sim, C.c_float), _ptr(li, C.c_int32), _ptr(ri, C.c_int32), C.byref(n)))
        if n.value < 0:
            return None
        return sim[:n.value], li[:n.value], ri[:n.value]

    def debug_lut(self, latent: FPTemplate) -> np.ndarray:
        v = _Views([latent])
        n = latent.tex[0].n if latent.tex else 0
        out = np.empty((max(n, 1), 16, 256), np.float32); nr = C.c_int32(0)
        self._chk(self._tap("afis_debug_lut")(self.ctx, v.arr, _ptr(out, C.c_float), C.byref(nr)))
        return out[:nr.value]

    def debug_texture_rowmax(self, latent: FPTemplate, g: int):
        v = _Views([latent])
        val = np.zeros(1000, np.float32); arg = np.zeros(1000, np.int32); nr = C.c_int32(0)
        self._chk(self._tap("afis_debug_texture_rowmax")(self.ctx, v.arr, g, _ptr(val, C.c_float), _ptr(arg, C.c_int32), C.byref(nr)))
        return val[:nr.value], arg[:nr.value]

    # ---- the reference's drivers (matching/matcher.cpp:96-337) ---------------------------------------------------
    def load_gallery_dir(self, rolled_path: str) -> List[str]:
        """Directory scan for *.dat as One2List/List2List do (matcher.cpp:120-130, directory order)."""
        if os.path.isfile(rolled_path):                   # a packed gallery container instead of a directory
            self.gallery_load(rolled_path)
            self.gallery_files = self.gallery_file_names(rolled_path)
            self.gallery_commit(0)
            return self.gallery_files
        files = [os.path.join(rolled_path, f) for f in os.listdir(rolled_path) if os.path.splitext(f)[1] == ".dat"]
        for f in files:
            with open(f, "rb") as fh:
                self.gallery_add_dat(fh.read())
        self.gallery_files = files
        self.gallery_commit(0)
        return files

    def One2List_matching(self, latent_template_file: str, score_path: str, top: int = 24) -> int:
        """matcher.cpp:216-337: rank list of the top 24 as `<rank>"<path>",<score>` under a `filename,score` header."""
        with open(latent_template_file, "rb") as f:
            buf = f.read()
        stem = os.path.splitext(os.path.basename(latent_template_file))[0]
        _, parsed = read_latent(buf)
        if not parsed.minu and not parsed.tex:                # matcher.cpp:260-268: a latent without any template gets a score file holding `0`
            with open(score_path + stem + ".csv", "w") as out:
                out.write("0\n")
        r = self.search_dat([buf], k=min(top, max(1, self.gallery_size)))
        if r["status"][0] == 1:
            return 1
        k = min(top, self.gallery_size)
        idx = [int(r["topk_idx"][0, j]) for j in range(k)]; top_sc = [r["topk_score"][0, j] for j in range(k)]
        if self.get_option("ref_tie_order") >= 1:             # the reference binary's own order of equal scores (std::sort on the score column), as `match -l -tie`
            ri, rs = self.rank_list(r["scores"][0], k, ref_order=True)
            idx = [int(g) for g in ri]; top_sc = list(rs)
        with open(score_path + stem + ".csv", "w") as out:
            out.write("filename,score\n")
            for j, g in enumerate(idx):
                out.write(f'{j + 1}"{self.gallery_files[g]}",{_cxx_float(top_sc[j])}\n')
        # correspondence files of the ranked templates (matcher.cpp:311-328, :497-505); the reference's hard-coded
        # /LatentAFIS/scores/ prefix becomes the score directory, as in the `match` CLI
        _, latent = read_latent(buf)
        for g, lists in zip(idx, self.correspondences(latent, idx)):
            rstem = os.path.splitext(os.path.basename(self.gallery_files[g]))[0]
            for i, xy in enumerate(lists):
                if xy is None:
                    continue
                with open(f"{score_path}corr{stem}_{rstem}_{i}.csv", "w") as cf:
                    for lx, ly, rx, ry in xy:
                        cf.write(f"{lx},{ly},{rx},{ry}\n")
        return 0

    def List2List_matching(self, latent_path: str, score_path: str) -> int:
        """matcher.cpp:96-214: one CSV per latent, one `"path",%.3f` line per gallery file."""
        files = [os.path.join(latent_path, f) for f in os.listdir(latent_path) if os.path.splitext(f)[1] == ".dat"]
        if not files:
            return -1
        bufs = [open(f, "rb").read() for f in files]
        r = self.search_dat(bufs, k=0)
        for i, f in enumerate(files):
            stem = os.path.splitext(os.path.basename(f))[0]
            _, parsed = read_latent(bufs[i])
            if not parsed.minu and not parsed.tex:            # matcher.cpp:153-163: `0` and on to the next latent
                with open(score_path + stem + ".csv", "w") as out:
                    out.write("0\n")
                continue
            if r["status"][i] == 1:
                continue
            with open(score_path + stem + ".csv", "w") as out:
                for j, gf in enumerate(self.gallery_files):
                    out.write(f'"{gf}",{r["scores"][i, j]:.3f}\n')
        return 0


def _cxx_float(v: float) -> str:
    """`ostream << float` with default formatting (6 significant digits, %g style)."""
    return "%g" % float(v)
