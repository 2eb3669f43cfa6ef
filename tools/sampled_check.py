#!/usr/bin/env python3
"""A frame rendered once (gmlrt_render, no costs known) with the sampled hand-out (GMLRT_SAMPLED=1) and without it: the
images must be the same byte for byte and the ray counts equal; prints the kernel and wall times (medians of five; a
small other scene is rendered in between so that the library keeps nothing of the frame).
Usage on the GPU box: python tools/sampled_check.py [scene ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np
import gmlrt

CONFIGS = {"spheres": (3840, 2160, None), "chess": (3840, 2160, 5), "fractal": (3840, 2160, None), "cone-fractal": (3840, 2160, None),
           "spotlight": (3840, 2160, None), "house": (3840, 2160, None), "large": (7680, 4320, 8)}


def main():
    names = sys.argv[1:] or ["chess", "fractal", "cone-fractal"]
    other = gmlrt.Program(path=os.path.join(ROOT, "scenes", "spheres.gml")).calls[0]
    gmlrt.init(1)
    bad = 0
    for name in names:
        w, h, depth = CONFIGS[name]
        call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(w, h, depth)
        gmlrt.render(call)
        res = {}
        for mode in ("1", "0", "1", "0"):
            os.environ["GMLRT_SAMPLED"] = mode
            kern, wall = res.setdefault(mode, ([], [], []))[:2]
            for _ in range(5):
                gmlrt.render(other)
                t = time.perf_counter()
                img, _, st = gmlrt.render(call)
                wall.append((time.perf_counter() - t) * 1e3)
                kern.append(st.kernel_ms)
            res[mode][2][:] = [img, (st.rays_primary, st.rays_shadow, st.rays_reflect, st.hits), st.handout_way]
        med = lambda v: sorted(v)[len(v) // 2]
        a, b = res["1"], res["0"]
        same = bool((a[2][0] == b[2][0]).all()) and a[2][1] == b[2][1]
        bad += 0 if same else 1
        print("%-13s sampled (way %d): kernel %.3f ms (min %.3f), wall %.3f ms | plain (way %d): kernel %.3f ms (min %.3f), wall %.3f ms | %s" % (
            name, a[2][2], med(a[0]), min(a[0]), med(a[1]), b[2][2], med(b[0]), min(b[0]), med(b[1]),
            "same image and counts" if same else "DIFFERENT: pixels %d, counts %s %s" % (int((a[2][0] != b[2][0]).any(-1).sum()), a[2][1], b[2][1])), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
