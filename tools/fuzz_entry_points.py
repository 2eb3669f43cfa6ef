#!/usr/bin/env python3
"""Parity fuzz of the other entry points: seeded random scenes (tests/random_scenes.py) rendered through the scene
handle (gmlrt_render_device) six times in a row (every way of handing out tiles a handle goes through), through the
entry point that allocates nothing, and as three interleaved row partitions, each against the CPU oracle's image and
ray counts.  Usage on the GPU box: python tools/fuzz_entry_points.py <number of seeds> [first seed] [width height depth]
(400 seeds from 5000 on at 257x193 depth 4: no difference, the last time it ran)."""
import ctypes, os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'rust-raytracer_amd')
import torch, numpy as np, random_scenes, oracle_py, gmlrt
N = int(sys.argv[1])
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
W, H, DEPTH = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (257, 193, 4)
bad = 0
for seed in range(FIRST, FIRST + N):
    t = random_scenes.scene_text(seed, W, H, DEPTH, lattice=(seed % 3 == 0))
    call = gmlrt.Program(text=t).calls[0]
    ref, _, cnt = oracle_py.OracleProgram(text=t).render(0, threads=16)
    want = (cnt["primary"], cnt["shadow"], cnt["reflect"])
    scene = gmlrt.DeviceScene(call.blob, 0)
    buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
    what = []
    for launch in range(6):
        buf.fill_(0x5A)
        st = scene.render_into(call.params, buf.data_ptr())
        if (buf.cpu().numpy() != ref).any() or (st.rays_primary, st.rays_shadow, st.rays_reflect) != want:
            what.append("handle launch %d (way %d)" % (launch, st.handout_way))
    buf.fill_(0x5A)
    st = scene.render_into(call.params, buf.data_ptr(), no_alloc=True)
    if (buf.cpu().numpy() != ref).any() or (st.rays_primary, st.rays_shadow, st.rays_reflect) != want:
        what.append("no-alloc")
    out = np.zeros_like(ref)
    rays = 0
    for part in range(3):
        p = gmlrt.Params()
        ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
        p.tile_rows, p.part, p.nparts = 8, part, 3
        rows = gmlrt.part_row_index(H, 8, part, 3)
        pb = torch.zeros((len(rows), W, 3), dtype=torch.uint8, device="cuda:0")
        st = scene.render_into(p, pb.data_ptr())
        out[rows] = pb.cpu().numpy()
        rays += st.rays
    if (out != ref).any() or rays != sum(want):
        what.append("three partitions")
    scene.close()
    if what:
        bad += 1
        print("seed", seed, "differs:", ", ".join(what), flush=True)
    if seed % 100 == 0:
        print("at", seed, flush=True)
print("done, bad =", bad)
