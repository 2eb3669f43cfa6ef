#!/usr/bin/env python3
"""Renders one configuration scene a few times (for `rocprofv3 --kernel-trace`: per-launch times of the level-by-level path).
Usage: python3 tools/levels_trace.py chess [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import gmlrt
from levels_check import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "chess"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h, depth = CONFIGS[name]
call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(w, h, depth)
scene = gmlrt.DeviceScene(call.blob, 0)
out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(n):
    st = scene.render_into(call.params, out.data_ptr())
    print("kernel_ms %.3f way %d rays %d" % (st.kernel_ms, st.handout_way, st.rays))
scene.close()
