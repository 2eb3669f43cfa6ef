#!/usr/bin/env python3
"""Host overhead of one gmlrt_render_device call (wall time per call minus its kernel time), for a long, a short and a
partitioned launch.  Usage on the GPU box: python tools/launch_overhead.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt, rowtiles
for name, w, h, d, nparts in (("chess", 3840, 2160, 5, 8), ("large", 1920, 1080, 8, 1), ("chess", 3840, 2160, 5, 1)):
    call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(w, h, d)
    scene = gmlrt.DeviceScene(call.blob, 0)
    p = gmlrt.Params()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
    p.tile_rows, p.part, p.nparts = 8, 0, nparts
    out = torch.zeros((rowtiles.padded_rows(h, 8, nparts), w, 3), dtype=torch.uint8, device="cuda:0")
    for _ in range(8):
        scene.render_into(p, out.data_ptr())
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    k = 0.0
    for _ in range(n):
        k += scene.render_into(p, out.data_ptr()).kernel_ms
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"{name} {w}x{h} part 0/{nparts}: wall/step {wall:.4f} ms, kernel {k / n:.4f} ms, overhead {1e3 * (wall - k / n):.1f} us")
    scene.close()
