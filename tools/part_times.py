#!/usr/bin/env python3
"""Kernel times of successive launches of one partition of the headline frame through a scene handle (the ways of handing
out tiles it goes through).  Usage on the GPU box: python tools/part_times.py <part> <nparts> [launches]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt, rowtiles
part, n = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
scene = gmlrt.DeviceScene(call.blob, 0)
p = gmlrt.Params()
ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
p.tile_rows, p.part, p.nparts = 8, part, n
out = torch.zeros((rowtiles.padded_rows(2160, 8, n), 3840, 3), dtype=torch.uint8, device="cuda:0")
sts = [scene.render_into(p, out.data_ptr()) for _ in range(k)]
print(" ".join("%.3f(w%d,s%d)" % (s.kernel_ms, s.handout_way, s.tiles_shared) for s in sts))
scene.close()
