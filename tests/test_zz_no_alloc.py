"""gmlrt_render_device_noalloc (the entry point that allocates nothing: its event stacks, context records and
counters live in a static __device__ array) against gmlrt_render_device, through the C ABI.

The bar is the one of the other GPU tests for this path, at its tight end: the two entry points run the same
kernel on the same rays, so bytes, unquantised f64 pixels (bit for bit) and ray counts must be equal.
Two calls in a row share the device's array; the second is the smaller one and must not see what the first left.
"""
import ctypes
import os
import struct

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu


def _render(gmlrt, scene, call, no_alloc):
    import torch
    h, w = call.params.height, call.params.width
    rgb8 = torch.full((h, w, 3), 77, dtype=torch.uint8, device="cuda:0")
    f64 = torch.full((h, w, 3), -1.0, dtype=torch.float64, device="cuda:0")
    p = gmlrt.Params()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
    p.tile_rows, p.part, p.nparts = 8, 0, 1
    st = scene.render_into(p, rgb8.data_ptr(), f64.data_ptr(), None, no_alloc=no_alloc)
    torch.cuda.synchronize()
    return rgb8.cpu().numpy(), f64.cpu().numpy(), st


def test_noalloc_entry_point_matches_render_device_on_two_calls_in_a_row(gmlrt):
    # first a deep CSG scene at depth 5 (long event stacks, every bounce level's parked record in use),
    # then, straight after it on the same device, a smaller frame of a small scene
    calls = [gmlrt.Program(path=os.path.join(SCENES, "chess.gml")).calls[0].with_overrides(640, 360, 5),
             gmlrt.Program(path=os.path.join(SCENES, "spheres.gml")).calls[0].with_overrides(200, 120, 3)]
    scenes = [gmlrt.DeviceScene(c.blob, 0) for c in calls]
    want = [_render(gmlrt, s, c, no_alloc=False) for s, c in zip(scenes, calls)]
    got = [_render(gmlrt, s, c, no_alloc=True) for s, c in zip(scenes, calls)]  # back to back, big then small
    for (img_w, f_w, st_w), (img_g, f_g, st_g), c in zip(want, got, calls):
        assert (img_g == img_w).all()
        assert (f_g.view(np.uint64) == f_w.view(np.uint64)).all()
        assert (st_g.rays_primary, st_g.rays_shadow, st_g.rays_reflect, st_g.hits) == \
               (st_w.rays_primary, st_w.rays_shadow, st_w.rays_reflect, st_w.hits)
        assert st_g.rays_primary == c.params.width * c.params.height
    # and once more in the other order: small, then big again
    again = [_render(gmlrt, scenes[i], calls[i], no_alloc=True) for i in (1, 0)]
    assert (again[0][0] == want[1][0]).all() and (again[1][0] == want[0][0]).all()
    assert again[1][2].rays == want[0][2].rays
    for s in scenes:
        s.close()


def test_noalloc_entry_point_renders_row_partitions(gmlrt):
    # the partition arguments mean the same as in gmlrt_render_device
    import torch
    call = gmlrt.Program(path=os.path.join(SCENES, "cone-fractal.gml")).calls[0].with_overrides(160, 100, 3)
    scene = gmlrt.DeviceScene(call.blob, 0)
    full = _render(gmlrt, scene, call, no_alloc=False)[0]
    h, w = call.params.height, call.params.width
    out = np.zeros_like(full)
    for part in range(3):
        p = gmlrt.Params()
        ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
        p.tile_rows, p.part, p.nparts = 8, part, 3
        rows = gmlrt.part_row_index(h, 8, part, 3)
        buf = torch.zeros((len(rows), w, 3), dtype=torch.uint8, device="cuda:0")
        scene.render_into(p, buf.data_ptr(), None, None, no_alloc=True)
        out[rows] = buf.cpu().numpy()
    assert (out == full).all()
    scene.close()


def test_noalloc_entry_point_rejects_what_its_array_cannot_hold(gmlrt):
    # a blob may declare more event slots than its tree needs (the validator only asks for "enough"):
    # 100000 slots of 12 bytes for each of a block's 256 rays exceed the 256 MiB array
    import torch
    call = gmlrt.Program(path=os.path.join(SCENES, "spheres.gml")).calls[0].with_overrides(64, 48, 1)
    blob = bytearray(call.blob)
    assert struct.unpack_from("<I", blob, 40)[0] >= 2  # max_events
    struct.pack_into("<I", blob, 40, 100000)
    scene = gmlrt.DeviceScene(bytes(blob), 0)
    buf = torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda:0")
    p = gmlrt.Params()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
    with pytest.raises(gmlrt.GmlrtError) as e:
        scene.render_into(p, buf.data_ptr(), None, None, no_alloc=True)
    assert e.value.status == -8 and "static scratch" in str(e.value)
    # the allocating entry point takes the same handle (it sizes its workspace to what the blob declares)
    st = scene.render_into(p, buf.data_ptr())
    assert st.rays_primary == 64 * 48
    scene.close()
