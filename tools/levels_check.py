#!/usr/bin/env python3
"""Level-by-level rendering (hand-out way 5) against the one-launch kernel on every configuration scene: the images must be the same byte for byte and the ray counts equal; prints the kernel times.
Usage on the GPU box: python tools/levels_check.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import gmlrt

CONFIGS = {"spheres": (3840, 2160, None), "chess": (3840, 2160, 5), "fractal": (3840, 2160, None), "cone-fractal": (3840, 2160, None),
           "spotlight": (3840, 2160, None), "house": (3840, 2160, None), "large": (7680, 4320, 8)}
MODES = ("plain", "levels")  # the one-launch kernel; one launch per bounce level


def main():
    names = sys.argv[1:] or list(CONFIGS)
    bad = 0
    for name in names:
        w, h, depth = CONFIGS[name]
        call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(w, h, depth)
        res = {}
        for mode in MODES:
            os.environ["GMLRT_LEVELS"] = "1" if mode == "levels" else "0"
            scene = gmlrt.DeviceScene(call.blob, 0)
            out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            sts = [scene.render_into(call.params, out.data_ptr()) for _ in range(8)]
            torch.cuda.synchronize()
            res[mode] = (out.cpu(), sts)
            scene.close()
        a, sa = res["plain"]
        line = "%-13s rays %d" % (name, sa[-1].rays)
        for mode in MODES:
            b, sb = res[mode]
            same = bool((a == b).all())
            counts = [(s.rays_primary, s.rays_shadow, s.rays_reflect, s.hits) for s in (sa[-1], sb[-1])]
            ok = same and counts[0] == counts[1]
            bad += 0 if ok else 1
            line += " | %s: first %.3f ms, warm %.3f ms (way %d)%s" % (mode, sb[0].kernel_ms, min(s.kernel_ms for s in sb[5:]), sb[-1].handout_way,
                    "" if ok else " DIFFERENT: pixels %d, counts %s" % (int((a != b).any(-1).sum()), counts))
        print(line, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
