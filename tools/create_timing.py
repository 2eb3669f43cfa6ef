#!/usr/bin/env python3
"""Time of gmlrt_scene_create (validation, derivation of the traversal and surface programs, uploads) for three scenes;
with GMLRT_VERBOSE=1 the derivation alone is logged too.  Usage on the GPU box: python tools/create_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import gmlrt
for name in ("chess", "fractal", "cone-fractal"):
    call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(3840, 2160, None)
    for i in range(3):
        t0 = time.perf_counter()
        sc = gmlrt.DeviceScene(call.blob, 0)
        t1 = time.perf_counter()
        sc.close()
        print(name, "blob bytes", len(call.blob), "scene create %.3f ms" % ((t1 - t0) * 1e3), flush=True)
