"""ctypes loader for the parity checker (oracle/).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def _load(path):
    return C.CDLL(path)


class Oracle:
    def __init__(self):
        so = os.path.join(ORACLE_DIR, "libafis_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, "afis_oracle.cpp")):
            build()
        L = self.lib = _load(so)
        vp, cp, ip, fp = C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.orc_codebook_from_bytes.restype = vp; L.orc_codebook_from_bytes.argtypes = [cp, C.c_long]
        L.orc_codebook_free.argtypes = [vp]
        L.orc_codebook_table_dist.restype = fp; L.orc_codebook_table_dist.argtypes = [vp]
        L.orc_latent_from_bytes.restype = vp; L.orc_latent_from_bytes.argtypes = [vp, cp, C.c_long, ip]
        L.orc_rolled_from_bytes.restype = vp; L.orc_rolled_from_bytes.argtypes = [cp, C.c_long, ip]
        L.orc_latent_free.argtypes = [vp]; L.orc_rolled_free.argtypes = [vp]
        L.orc_latent_counts.argtypes = [vp, ip]; L.orc_rolled_counts.argtypes = [vp, ip]
        for f in (L.orc_latent_minu_n, L.orc_latent_tex_n, L.orc_rolled_minu_n, L.orc_rolled_tex_n):
            f.restype = C.c_int; f.argtypes = [vp, C.c_int]
        L.orc_latent_lut.restype = fp; L.orc_latent_lut.argtypes = [vp, C.c_int]
        L.orc_rolled_codes.restype = C.POINTER(C.c_uint8); L.orc_rolled_codes.argtypes = [vp, C.c_int]
        L.orc_points.restype = C.c_int; L.orc_points.argtypes = [vp, C.c_int, C.c_int, ip, ip, fp]
        L.orc_build_lut.argtypes = [vp, fp, C.c_int, C.c_int, fp]
        L.orc_all_templates.restype = C.c_int; L.orc_all_templates.argtypes = [vp, vp, vp, C.c_int, fp]
        L.orc_pq_encode.restype = None; L.orc_pq_encode.argtypes = [vp, fp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
        L.orc_pair_score.restype = C.c_int; L.orc_pair_score.argtypes = [vp, vp, vp, C.c_int, fp]
        L.orc_texture_rowmax.restype = C.c_int; L.orc_texture_rowmax.argtypes = [vp, vp, vp, fp, ip]
        L.orc_trace.restype = C.c_int; L.orc_trace.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, fp, ip, ip]
        L.orc_search.restype = C.c_int
        L.orc_search.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_num_threads.restype = C.c_int
        L.orc_search_files.restype = C.c_int
        L.orc_search_files.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, fp]
        L.orc_atan2f_grid.restype = None; L.orc_atan2f_grid.argtypes = [C.c_int, fp]

    # -- handles
    def codebook(self, buf: bytes):
        h = self.lib.orc_codebook_from_bytes(buf, len(buf))
        if not h:
            raise ValueError("codebook is empty!")
        return h

    def latent(self, cb, buf: bytes):
        rc = C.c_int(0)
        h = self.lib.orc_latent_from_bytes(cb, buf, len(buf), C.byref(rc))
        return h, rc.value

    def rolled(self, buf: bytes):
        rc = C.c_int(0)
        h = self.lib.orc_rolled_from_bytes(buf, len(buf), C.byref(rc))
        return h, rc.value

    def counts(self, h, rolled=False):
        c = (C.c_int * 2)()
        (self.lib.orc_rolled_counts if rolled else self.lib.orc_latent_counts)(h, c)
        return c[0], c[1]

    def lut(self, lat, t, M=16, K=256):
        n = self.lib.orc_latent_tex_n(lat, t)
        p = self.lib.orc_latent_lut(lat, t)
        return np.ctypeslib.as_array(p, shape=(n, M, K)).copy()

    def build_lut(self, cb, des, M=16, K=256):
        des = np.ascontiguousarray(des, dtype=np.float32)
        out = np.empty((des.shape[0], M, K), dtype=np.float32)
        self.lib.orc_build_lut(cb, des.ctypes.data_as(C.POINTER(C.c_float)), des.shape[0], des.shape[1],
                               out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def pair(self, cb, lat, rol, tie_mode=1):
        out = (C.c_float * 5)()
        rc = self.lib.orc_pair_score(cb, lat, rol, tie_mode, out)
        return rc, np.array(out[:], dtype=np.float32)

    def texture_rowmax(self, cb, lat, rol):
        val = np.zeros(1000, np.float32); arg = np.zeros(1000, np.int32)
        n = self.lib.orc_texture_rowmax(cb, lat, rol, val.ctypes.data_as(C.POINTER(C.c_float)), arg.ctypes.data_as(C.POINTER(C.c_int)))
        return val[:n], arg[:n]

    def pq_encode(self, cb, des):
        des = np.ascontiguousarray(des, np.float32)
        codes = np.zeros((des.shape[0], 16), np.uint8)
        self.lib.orc_pq_encode(cb, des.ctypes.data_as(C.POINTER(C.c_float)), des.shape[0], des.shape[1], codes.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return codes

    def all_templates(self, cb, lat, rol, width, tie_mode=1):
        out = np.zeros(max(1, width), np.float32)
        rc = self.lib.orc_all_templates(cb, lat, rol, tie_mode, out.ctypes.data_as(C.POINTER(C.c_float)))
        return rc, out[:width]

    def trace(self, cb, lat, rol, which, stage, tie_mode=1):
        sim = np.zeros(256, np.float32); li = np.zeros(256, np.int32); ri = np.zeros(256, np.int32)
        n = self.lib.orc_trace(cb, lat, rol, tie_mode, which, stage, sim.ctypes.data_as(C.POINTER(C.c_float)),
                               li.ctypes.data_as(C.POINTER(C.c_int)), ri.ctypes.data_as(C.POINTER(C.c_int)))
        if n < 0:
            return None
        return sim[:n], li[:n], ri[:n]

    def search_files(self, cb, lat, paths, tie_mode=1, threads=0):
        """The reference's own loop: every rolled .dat re-read and re-parsed per pair (matcher.cpp:173, :278)."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        scores = np.empty(n, np.float32)
        rc = self.lib.orc_search_files(cb, lat, arr, n, tie_mode, threads, scores.ctypes.data_as(C.POINTER(C.c_float)))
        return rc, scores

    def atan2f_grid(self, R):
        out = np.empty((2 * R + 1, 2 * R + 1), np.float32)
        self.lib.orc_atan2f_grid(R, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def rank_list(self, scores, std_sort: bool):
        """matcher.cpp:306-309: gallery indices by descending score; std_sort: libstdc++'s std::sort as the reference (equal scores in introsort's order), else ascending index."""
        s = np.ascontiguousarray(scores, np.float32); out = np.empty(len(s), np.int32)
        self.lib.orc_rank_list.restype = None
        self.lib.orc_rank_list(s.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(len(s)), C.c_int(int(std_sort)), out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out

    def search(self, cb, lat, rolled_handles, tie_mode=1, threads=0, want_parts=False):
        n = len(rolled_handles)
        arr = (C.c_void_p * n)(*rolled_handles)
        scores = np.empty(n, np.float32)
        parts = np.zeros((n, 5), np.float32) if want_parts else None
        rc = self.lib.orc_search(cb, lat, arr, n, tie_mode, threads, scores.ctypes.data_as(C.POINTER(C.c_float)),
                                 parts.ctypes.data_as(C.POINTER(C.c_float)) if want_parts else None)
        return (rc, scores, parts) if want_parts else (rc, scores)


class RefHarness:
    """oracle/_ref/libafis_ref.so — compiled from the reference's own include.h (this container only builds it)."""

    def __init__(self):
        so = os.path.join(ORACLE_DIR, "_ref", "libafis_ref.so")
        if not os.path.exists(so):
            build()
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        self.lib = _load(so)
        self.lib.ref_pi.restype = C.c_double

    def arg(self, argv, opt):
        """(exists, value) of `opt` in the command line `argv` (argv[0] = program name), per matching/argparser.h."""
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        out = C.create_string_buffer(4096)
        ex = self.lib.ref_arg(len(argv), arr, opt.encode(), out, 4096)
        return bool(ex), out.value.decode()

    def config_get(self, path, key):
        """(rc, value): main.cpp:41-44 with the JSON library vendored in the reference (1 found, 0 absent, -1 unreadable)."""
        out = C.create_string_buffer(8192)
        rc = self.lib.ref_config_get(path.encode(), key.encode(), out, 8192)
        return rc, out.value.decode()

    def build_lut(self, des, words):
        des = np.ascontiguousarray(des, np.float32); words = np.ascontiguousarray(words, np.float32)
        n, dl = des.shape; M, K, dsub = words.shape
        x = np.zeros(n, np.int16); y = np.zeros(n, np.int16); ori = np.zeros(n, np.float32)
        out = np.empty((n, M, K), np.float32)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self.lib.ref_build_lut(n, p(x, C.c_short), p(y, C.c_short), p(ori, C.c_float), dl, p(des, C.c_float),
                               p(words, C.c_float), M, dsub, K, p(out, C.c_float))
        return out

    def rolled_texture(self, x, y, ori, des_len, raw_bytes: bytes):
        n = len(x)
        x = np.ascontiguousarray(x, np.int16); y = np.ascontiguousarray(y, np.int16); ori = np.ascontiguousarray(ori, np.float32)
        buf = np.zeros(n * des_len * 4, np.uint8)
        buf[:len(raw_bytes)] = np.frombuffer(raw_bytes, np.uint8)[:len(buf)]
        codes = np.empty((n, des_len), np.uint8); xo = np.empty(n, np.int32); yo = np.empty(n, np.int32); oo = np.empty(n, np.float32)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self.lib.ref_rolled_texture(n, p(x, C.c_short), p(y, C.c_short), p(ori, C.c_float), des_len, p(buf, C.c_float),
                                    p(codes, C.c_uint8), p(xo, C.c_int), p(yo, C.c_int), p(oo, C.c_float))
        return codes, xo, yo, oo
