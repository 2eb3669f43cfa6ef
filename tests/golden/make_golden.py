#!/usr/bin/env python3
"""Regenerates tests/golden/*.npy and golden.json with the CPU oracle (oracle/liboracle.so).

These are regression fixtures of THIS repository's oracle: the seven configuration scenes at a reduced size.
They are not reference output — the reference cannot be built here (no Rust toolchain) and holds no images.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402

SIZES = {"spheres": (96, 72, 3), "spotlight": (96, 72, 1), "house": (100, 60, 3), "cone-fractal": (96, 64, 3),
         "chess": (128, 96, 5), "fractal": (96, 64, 3), "large": (96, 60, 1)}

meta = {}
for name, (w, h, depth) in SIZES.items():
    prog = oracle_py.OracleProgram(path=os.path.join(ROOT, "scenes", name + ".gml"))
    img, _, cnt = prog.render(0, w=w, h=h, depth=depth, threads=8)
    np.save(os.path.join(HERE, name + ".npy"), img)
    meta[name] = {"w": w, "h": h, "depth": depth, "counts": cnt}
    print(name, w, h, depth, cnt)
json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
