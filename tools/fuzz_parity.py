#!/usr/bin/env python3
"""Parity fuzz: seeded random scenes (tests/random_scenes.py) at 160x120, depth 4, on the GPU against the CPU oracle;
prints every seed whose image or ray counts differ.  Usage on the GPU box:
python tools/fuzz_parity.py <number of seeds> [lattice|lattice45|plain] [first seed] [width height depth]
(5000 plain + 2000 lattice seeds from 1000 on and 500 seeds at 320x240 depth 6: no difference, the last time it ran)."""
import sys, os
ROOT=os.getcwd()
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'rust-raytracer_amd')
import torch, numpy as np, random_scenes, oracle_py, gmlrt
bad=0
LATTICE = (2 if sys.argv[2] == "lattice45" else sys.argv[2] == "lattice") if len(sys.argv) > 2 else False
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
W, H, DEPTH = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (161 if LATTICE else 160, 121 if LATTICE else 120, 4)
for seed in range(FIRST, FIRST+int(sys.argv[1])):
    t = random_scenes.scene_text(seed, W, H, DEPTH, lattice=LATTICE)
    try:
        img, f64, st = gmlrt.render(gmlrt.Program(text=t).calls[0], want_f64=True)
    except Exception as e:
        print("seed", seed, "GPU ERR", e, flush=True); bad+=1; continue
    ref, ref64, cnt = oracle_py.OracleProgram(text=t).render(0, want_f64=True, threads=16)
    nd = int((img != ref).any(axis=-1).sum())
    rc = (st.rays_primary, st.rays_shadow, st.rays_reflect) == (cnt["primary"], cnt["shadow"], cnt["reflect"])
    if nd or not rc:
        bad+=1; print("seed", seed, "differing", nd, "rays equal", rc, flush=True)
    if seed % 100 == 0: print("at", seed, flush=True)
print("done, bad =", bad)
