#!/usr/bin/env python3
"""One partition of the headline frame with the diagnostic build (GMLRT_LIB=... GMLRT_VERBOSE=1): its dearest tiles and what they did.
Usage on the GPU box: python tools/part_diag.py <part> <nparts>"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt, rowtiles
part, n = int(sys.argv[1]), int(sys.argv[2])
call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
scene = gmlrt.DeviceScene(call.blob, 0)
p = gmlrt.Params()
ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
p.tile_rows, p.part, p.nparts = 8, part, n
out = torch.zeros((rowtiles.padded_rows(2160, 8, n), 3840, 3), dtype=torch.uint8, device="cuda:0")
st = scene.render_into(p, out.data_ptr())
print("kernel_ms", st.kernel_ms, "rays", st.rays)
scene.close()
