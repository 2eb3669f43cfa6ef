"""ctypes binding of libgmlrt.so — the C ABI in include/gmlrt.h.

Host-side mirror of the reference's public surface for this path:
`render_gml(text)` (lib.rs:28-30), the `rungml FILE` include handling
(rungml/src/main.rs:17-47) and the operands of `render`
(evaluator.rs:749-769).  There is no CPU fallback: if the HIP library is
missing or no GPU is visible, calls raise.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMLRT_LIB") or os.path.join(_HERE, "libgmlrt.so")  # GMLRT_LIB: try a variant build

# every symbol include/gmlrt.h declares
EXPORTS = [
    "gmlrt_strerror", "gmlrt_last_error", "gmlrt_part_rows", "gmlrt_part_row_index", "gmlrt_init",
    "gmlrt_shutdown", "gmlrt_device_count", "gmlrt_render", "gmlrt_scene_create", "gmlrt_scene_destroy",
    "gmlrt_render_device", "gmlrt_render_device_noalloc", "gmlrt_copy_rows_to_host", "gmlrt_render_to_sink", "gmlrt_write_ppm", "gmlrt_program_open_file", "gmlrt_program_open_text",
    "gmlrt_program_close", "gmlrt_program_num_renders", "gmlrt_program_render_call",
]


class Params(ctypes.Structure):
    _fields_ = [("amb", ctypes.c_double * 3), ("depth", ctypes.c_int64), ("fov_deg", ctypes.c_double),
                ("width", ctypes.c_int64), ("height", ctypes.c_int64), ("tile_rows", ctypes.c_int32),
                ("part", ctypes.c_int32), ("nparts", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("rays_primary", ctypes.c_uint64), ("rays_shadow", ctypes.c_uint64), ("rays_reflect", ctypes.c_uint64),
                ("kernel_ms", ctypes.c_double), ("total_ms", ctypes.c_double), ("tiles_shared", ctypes.c_uint32), ("handout_way", ctypes.c_uint32),
                ("hits", ctypes.c_uint64), ("n_gpus", ctypes.c_uint32), ("surface_errors", ctypes.c_uint32)]

    def as_dict(self):
        return {"primary": self.rays_primary, "shadow": self.rays_shadow, "reflect": self.rays_reflect,
                "kernel_ms": self.kernel_ms, "total_ms": self.total_ms, "hits": self.hits,
                "n_gpus": self.n_gpus, "tiles_shared": self.tiles_shared, "handout_way": self.handout_way, "surface_errors": self.surface_errors}

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect


NEW_IMAGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64)
PUSH_PIXEL_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double))
DONE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class Renderer(ctypes.Structure):
    """gmlrt_renderer: the reference's trait Renderer as C callbacks."""
    _fields_ = [("user", ctypes.c_void_p), ("new_image", NEW_IMAGE_FN), ("push_pixel", PUSH_PIXEL_FN),
                ("done", DONE_FN)]


ERR_SURFACE = -11  # a surface function failed at run time; the outputs were delivered all the same


class GmlrtError(RuntimeError):
    def __init__(self, status, where, result=None):
        L = lib()
        super().__init__("%s: %s: %s" % (where, L.gmlrt_strerror(status).decode(), L.gmlrt_last_error().decode()))
        self.status = status
        self.result = result  # ERR_SURFACE: what the call delivered


_LIB = None


def build():
    """Compile libgmlrt.so for gfx950 with the committed Makefile (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "-j8"])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgmlrt.so is not built (run `make -C rust-raytracer_amd/csrc`); "
                               "there is no CPU fallback for the render path")
        # PyTorch ships its own HIP runtime under the same SONAME; a process must hold only one.  Loading torch
        # first makes this library bind to that copy, so torch tensors and torch.distributed work beside it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.gmlrt_strerror.restype = ctypes.c_char_p
        L.gmlrt_strerror.argtypes = [ctypes.c_int]
        L.gmlrt_last_error.restype = ctypes.c_char_p
        L.gmlrt_part_rows.restype = ctypes.c_int64
        L.gmlrt_part_rows.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.gmlrt_part_row_index.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
        L.gmlrt_init.argtypes = [ctypes.c_int]
        L.gmlrt_render.argtypes = [vp, sz, ctypes.POINTER(Params), vp, vp, ctypes.POINTER(Stats)]
        L.gmlrt_render_to_sink.argtypes = [vp, sz, ctypes.POINTER(Params), ctypes.c_char_p, ctypes.POINTER(Renderer),
                                           ctypes.POINTER(Stats)]
        L.gmlrt_scene_create.argtypes = [vp, sz, ctypes.c_int, ctypes.POINTER(vp)]
        L.gmlrt_scene_destroy.argtypes = [vp]
        L.gmlrt_scene_destroy.restype = None
        L.gmlrt_render_device.argtypes = [vp, ctypes.POINTER(Params), vp, vp, vp, ctypes.POINTER(Stats)]
        L.gmlrt_render_device_noalloc.argtypes = L.gmlrt_render_device.argtypes
        L.gmlrt_copy_rows_to_host.argtypes = [ctypes.c_int, vp, vp, sz, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.gmlrt_write_ppm.argtypes = [ctypes.c_char_p, vp, ctypes.c_int64, ctypes.c_int64]
        L.gmlrt_program_open_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        L.gmlrt_program_open_text.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        L.gmlrt_program_close.argtypes = [vp]
        L.gmlrt_program_close.restype = None
        L.gmlrt_program_num_renders.argtypes = [vp]
        L.gmlrt_program_render_call.argtypes = [vp, ctypes.c_int, ctypes.POINTER(Params), ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.POINTER(vp), ctypes.POINTER(sz)]
        _LIB = L
    return _LIB


def _check(status, where):
    if status != 0:
        raise GmlrtError(status, where)


def part_rows(height, tile_rows, part, nparts):
    return lib().gmlrt_part_rows(height, tile_rows, part, nparts)


def part_row_index(height, tile_rows, part, nparts):
    n = part_rows(height, tile_rows, part, nparts)
    out = np.zeros(n, dtype=np.int64)
    _check(lib().gmlrt_part_row_index(height, tile_rows, part, nparts, out.ctypes.data_as(ctypes.c_void_p)),
           "gmlrt_part_row_index")
    return out


class RenderCall:
    """Operands of one executed `render` op: params, output file name and the flattened scene blob."""

    def __init__(self, params, filename, blob):
        self.params = params
        self.filename = filename
        self.blob = blob  # bytes

    def with_overrides(self, width=None, height=None, depth=None):
        p = Params()
        ctypes.memmove(ctypes.byref(p), ctypes.byref(self.params), ctypes.sizeof(Params))
        if width:
            p.width = width
        if height:
            p.height = height
        if depth is not None:
            p.depth = depth
        return RenderCall(p, self.filename, self.blob)


class Program:
    """An evaluated GML program (host front-end): the list of its render calls."""

    def __init__(self, path=None, text=None):
        h = ctypes.c_void_p()
        if path is not None:
            _check(lib().gmlrt_program_open_file(os.fsencode(path), ctypes.byref(h)), "gmlrt_program_open_file")
        else:
            _check(lib().gmlrt_program_open_text(text.encode(), ctypes.byref(h)), "gmlrt_program_open_text")
        self.calls = []
        try:
            for i in range(lib().gmlrt_program_num_renders(h)):
                p = Params()
                name = ctypes.c_char_p()
                blob = ctypes.c_void_p()
                n = ctypes.c_size_t()
                _check(lib().gmlrt_program_render_call(h, i, ctypes.byref(p), ctypes.byref(name), ctypes.byref(blob),
                                                       ctypes.byref(n)), "gmlrt_program_render_call")
                self.calls.append(RenderCall(p, name.value.decode(), ctypes.string_at(blob.value, n.value)))
        finally:
            lib().gmlrt_program_close(h)


_inited = 0


def init(n_gpus=0):
    global _inited
    _check(lib().gmlrt_init(n_gpus), "gmlrt_init")
    _inited = n_gpus or lib().gmlrt_device_count()
    return _inited


def render(call, want_f64=False):
    """Host-buffer drop-in (gmlrt_render): returns (rgb8 [h,w,3], f64 or None, Stats)."""
    if not _inited:
        init(0)
    p = call.params
    rgb8 = np.zeros((p.height, p.width, 3), dtype=np.uint8)
    f64 = np.zeros((p.height, p.width, 3), dtype=np.float64) if want_f64 else None
    st = Stats()
    rc = lib().gmlrt_render(call.blob, len(call.blob), ctypes.byref(p), rgb8.ctypes.data_as(ctypes.c_void_p),
                            f64.ctypes.data_as(ctypes.c_void_p) if want_f64 else None, ctypes.byref(st))
    if rc == ERR_SURFACE:
        raise GmlrtError(rc, "gmlrt_render", (rgb8, f64, st))
    _check(rc, "gmlrt_render")
    return rgb8, f64, st


def render_to_sink(call, new_image, push_pixel, done=None):
    """gmlrt_render_to_sink with Python callables: new_image(filename, w, h), push_pixel((r, g, b)), done();
    a callable that returns a true value stops the render (GMLRT_ERR_SINK)."""
    if not _inited:
        init(0)
    r = Renderer(None, NEW_IMAGE_FN(lambda u, f, w, h: int(bool(new_image(f.decode(), w, h)))),
                 PUSH_PIXEL_FN(lambda u, p: int(bool(push_pixel((p[0], p[1], p[2]))))),
                 DONE_FN(lambda u: int(bool(done()))) if done else DONE_FN())
    st = Stats()
    _check(lib().gmlrt_render_to_sink(call.blob, len(call.blob), ctypes.byref(call.params), call.filename.encode(),
                                      ctypes.byref(r), ctypes.byref(st)), "gmlrt_render_to_sink")
    return st


class DeviceScene:
    """A scene resident on one GPU (gmlrt_scene_create / gmlrt_render_device)."""

    def __init__(self, blob, device=0):
        self.h = ctypes.c_void_p()
        self._blob = blob
        _check(lib().gmlrt_scene_create(blob, len(blob), device, ctypes.byref(self.h)), "gmlrt_scene_create")

    def render_into(self, params, d_rgb8_ptr, d_f64_ptr=None, stream=None, no_alloc=False):
        """gmlrt_render_device, or with no_alloc its variant that allocates nothing (static scratch array)."""
        st = Stats()
        fn = lib().gmlrt_render_device_noalloc if no_alloc else lib().gmlrt_render_device
        _check(fn(self.h, ctypes.byref(params), d_rgb8_ptr, d_f64_ptr, stream, ctypes.byref(st)),
               "gmlrt_render_device_noalloc" if no_alloc else "gmlrt_render_device")
        return st

    def close(self):
        if self.h:
            lib().gmlrt_scene_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def copy_rows_to_host(device, d_rows_ptr, image, tile_rows, part, nparts):
    """gmlrt_copy_rows_to_host: the packed rows at device pointer d_rows_ptr go to their places in `image`, a
    C-contiguous numpy array [height, ...] in host memory (possibly a shared mapping)."""
    assert image.flags["C_CONTIGUOUS"]
    height = image.shape[0]
    row_bytes = image.strides[0]
    _check(lib().gmlrt_copy_rows_to_host(device, d_rows_ptr, image.ctypes.data_as(ctypes.c_void_p), row_bytes, height,
                                         tile_rows, part, nparts), "gmlrt_copy_rows_to_host")


def write_ppm(path, rgb8):
    h, w, _ = rgb8.shape
    a = np.ascontiguousarray(rgb8)
    _check(lib().gmlrt_write_ppm(os.fsencode(path), a.ctypes.data_as(ctypes.c_void_p), w, h), "gmlrt_write_ppm")
