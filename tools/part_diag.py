import ctypes, os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch, gmlrt, rowtiles
call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
scene = gmlrt.DeviceScene(call.blob, 0)
for n in (1, 8):
    p = gmlrt.Params()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
    p.tile_rows, p.part, p.nparts = 8, 0, n
    out = torch.zeros((rowtiles.padded_rows(2160, 8, n), 3840, 3), dtype=torch.uint8, device="cuda:0")
    print("N", n, file=sys.stderr)
    st = scene.render_into(p, out.data_ptr())
    print("kernel_ms", st.kernel_ms, file=sys.stderr)
