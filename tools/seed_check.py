#!/usr/bin/env python3
"""One seeded random scene (tests/random_scenes.py) on the GPU against the CPU oracle: the differing pixels, their
values on both sides and the ray counts.  Usage on the GPU box: python tools/seed_check.py <seed> [lattice|lattice45|plain] [w h depth]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "oracle", "rust-raytracer_amd"):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch, numpy as np, random_scenes, oracle_py, gmlrt
seed = int(sys.argv[1])
lattice = (2 if sys.argv[2] == "lattice45" else sys.argv[2] == "lattice") if len(sys.argv) > 2 else False
w, h, depth = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (161 if lattice else 160, 121 if lattice else 120, 4)
t = random_scenes.scene_text(seed, w, h, depth, lattice=lattice)
if os.environ.get("SEED_CHECK_TEXT"):
    print(t)
img, f64, st = gmlrt.render(gmlrt.Program(text=t).calls[0], want_f64=True)
ref, ref64, cnt = oracle_py.OracleProgram(text=t).render(0, want_f64=True, threads=16)
bad = np.argwhere((img != ref).any(axis=-1))
print("seed", seed, "differing pixels", len(bad), "rays", (st.rays_primary, st.rays_shadow, st.rays_reflect), "oracle", (cnt["primary"], cnt["shadow"], cnt["reflect"]))
for y, x in bad[:16]:
    print("  pixel x %d y %d: gpu %s oracle %s   f64 gpu %s oracle %s" % (x, y, img[y, x], ref[y, x], f64[y, x], ref64[y, x]))
