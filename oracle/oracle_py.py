"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped render path never loads it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """(Re)build liboracle.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_open.restype = ctypes.c_void_p
        L.oracle_open.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.oracle_open_text.restype = ctypes.c_void_p
        L.oracle_open_text.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.oracle_close.argtypes = [ctypes.c_void_p]
        L.oracle_num_renders.argtypes = [ctypes.c_void_p]
        L.oracle_render_params.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_double)]
        L.oracle_render_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_int]
        L.oracle_kat_intersect.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_double)] * 5 + [ctypes.c_int]
        L.oracle_kat_quantise.restype = ctypes.c_uint8
        L.oracle_kat_quantise.argtypes = [ctypes.c_double]
        _LIB = L
    return _LIB


class OracleProgram:
    """An evaluated GML program; each `render` op it executed can be rendered on the CPU."""

    def __init__(self, path=None, text=None):
        err = ctypes.create_string_buffer(512)
        if path is not None:
            self.h = lib().oracle_open(os.fsencode(path), err, 512)
        else:
            self.h = lib().oracle_open_text(text.encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())

    def close(self):
        if self.h:
            lib().oracle_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module's globals may be gone already
            pass

    @property
    def num_renders(self):
        return lib().oracle_num_renders(self.h)

    def params(self, idx=0):
        out = (ctypes.c_int64 * 3)()
        fov = ctypes.c_double()
        if lib().oracle_render_params(self.h, idx, out, ctypes.byref(fov)) != 0:
            raise IndexError(idx)
        return {"w": out[0], "h": out[1], "depth": out[2], "fov": fov.value}

    def render(self, idx=0, w=None, h=None, depth=None, y0=0, y1=None, ystep=1, threads=1, want_f64=False):
        """Returns (rgb8 [rows,w,3] uint8, f64 or None, counts dict)."""
        p = self.params(idx)
        w = p["w"] if w is None else w
        h = p["h"] if h is None else h
        depth = p["depth"] if depth is None else depth
        y1 = h if y1 is None else y1
        nrows = len(range(y0, y1, ystep))
        rgb8 = np.zeros((nrows, w, 3), dtype=np.uint8)
        f64 = np.zeros((nrows, w, 3), dtype=np.float64) if want_f64 else None
        counts = (ctypes.c_uint64 * 3)()
        err = ctypes.create_string_buffer(512)
        rc = lib().oracle_render_rows(self.h, idx, w, h, depth, y0, y1, ystep, threads,
                                      rgb8.ctypes.data_as(ctypes.c_void_p),
                                      f64.ctypes.data_as(ctypes.c_void_p) if want_f64 else None,
                                      counts, err, 512)
        if rc != 0:
            raise RuntimeError("oracle render: " + err.value.decode())
        return rgb8, f64, {"primary": counts[0], "shadow": counts[1], "reflect": counts[2]}


def kat_intersect(prim_type, translate, scale, raypos, raydir):
    D = ctypes.c_double * 3
    out = (ctypes.c_double * 12)()
    n = lib().oracle_kat_intersect(prim_type, D(*translate), D(*scale), D(*raypos), D(*raydir), out, 4)
    return [(bool(out[3 * i]), out[3 * i + 1], int(out[3 * i + 2])) for i in range(n)]


def quantise(v):
    return lib().oracle_kat_quantise(v)
