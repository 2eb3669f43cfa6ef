#!/usr/bin/env python3
"""Kernel time of ONE rank's share (part 0) of the headline frame at N = 4, 8 ranks for several grid sizes.
Usage on the GPU box: python tools/part_grid_sweep.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt, rowtiles

call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
for n in (1, 4, 8):
    for blocks in (1024, 768, 512, 384, 256):
        os.environ["GMLRT_GRID_BLOCKS"] = str(blocks)
        scene = gmlrt.DeviceScene(call.blob, 0)
        worst = 0.0
        for part in range(0, n, max(1, n // 2)):
            p = gmlrt.Params()
            ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
            p.tile_rows, p.part, p.nparts = 8, part, n
            out = torch.zeros((rowtiles.padded_rows(2160, 8, n), 3840, 3), dtype=torch.uint8, device="cuda:0")
            for _ in range(3):
                scene.render_into(p, out.data_ptr())
            ms = min(scene.render_into(p, out.data_ptr()).kernel_ms for _ in range(5))
            worst = max(worst, ms)
        scene.close()
        print(f"N={n} blocks={blocks}: kernel {worst:.3f} ms", flush=True)
