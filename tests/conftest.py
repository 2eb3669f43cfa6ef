import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, make_dir):
    # Built artefacts travel with the tree; build here only if one is missing and a compiler exists.
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", make_dir])
    return path


@pytest.fixture(scope="session")
def helpers():
    """Test-only native helpers (front-end + surface compiler exports)."""
    p = _ensure(os.path.join(ROOT, "tests", "native", "libgmlrt_testhelpers.so"), os.path.join(ROOT, "tests", "native"))
    L = ctypes.CDLL(p)
    L.th_surface_compare.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                                     ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
    L.th_surface_eval_text.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_double,
                                       ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
    L.th_transform.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double)]

    class H:
        lib = L

        @staticmethod
        def _s(fn, *args):
            buf = ctypes.create_string_buffer(1 << 16)
            fn(*args, buf, 1 << 16)
            return buf.value.decode()

        def tokenize(self, text):
            return self._s(L.th_tokenize, text.encode())

        def parse(self, text):
            return self._s(L.th_parse, text.encode())

        def run(self, text, var=""):
            return self._s(L.th_run, text.encode(), var.encode())

        def preprocess(self, path):
            return self._s(L.th_preprocess, os.fsencode(path))

    return H()


@pytest.fixture(scope="session")
def oracle():
    _ensure(os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "oracle"))
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def gmlrt():
    import gmlrt as g
    return g


def image_diff(a, b):
    """(fraction of pixels with every channel within 1, max abs channel difference, #pixels differing at all)."""
    import numpy as np
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    per_px = d.max(axis=-1)
    return float((per_px <= 1).mean()), int(per_px.max()), int((per_px > 0).sum())
