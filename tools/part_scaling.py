#!/usr/bin/env python3
"""Kernel time of one rank's share of the headline frame for N = 1, 2, 4, 8 ranks, measured on ONE GPU
(each rank's partition rendered alone).  Shows what strong scaling can reach before any gather cost.
Usage on the GPU box: python tools/part_scaling.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt, rowtiles

call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
scene = gmlrt.DeviceScene(call.blob, 0)
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    wall = 0.0
    for part in range(n):
        p = gmlrt.Params()
        ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
        p.tile_rows, p.part, p.nparts = 8, part, n
        out = torch.zeros((rowtiles.padded_rows(2160, 8, n), 3840, 3), dtype=torch.uint8, device="cuda:0")
        for _ in range(6):  # one launch that times the tiles, three ways of handing them out on trial (the third twice), one decided
            scene.render_into(p, out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms = 0.0
        for _ in range(5):
            ms += scene.render_into(p, out.data_ptr()).kernel_ms
        torch.cuda.synchronize()
        wall = max(wall, (time.perf_counter() - t0) / 5 * 1e3)
        worst = max(worst, ms / 5)
    base = base or worst
    print(f"N={n}: slowest rank kernel {worst:.3f} ms, wall/step {wall:.3f} ms, ideal {base / n:.3f} ms, efficiency {base / n / worst:.2f}")
