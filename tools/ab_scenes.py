#!/usr/bin/env python3
"""A/B of library builds on the configuration scenes: every build renders every scene in a process of its own (the
learning launches, then timed ones), the builds take turns for a number of rounds, and the table shows the best and
the median warm kernel time per build and scene, the first-launch time, and whether the images and ray counts of all
builds are the same (SHA-1 of the bytes).
Usage on the GPU box: python tools/ab_scenes.py [--rounds 2] [--scenes chess,fractal] base=path/to/libgmlrt.so look=rust-raytracer_amd/variants/libgmlrt_x.so ..."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"spheres": (3840, 2160, None), "chess": (3840, 2160, 5), "fractal": (3840, 2160, None), "cone-fractal": (3840, 2160, None),
           "spotlight": (3840, 2160, None), "house": (3840, 2160, None), "large": (7680, 4320, 8)}


def worker(names):
    sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
    import torch
    import gmlrt
    res = {}
    for name in names:
        w, h, depth = CONFIGS[name]
        call = gmlrt.Program(path=os.path.join(ROOT, "scenes", name + ".gml")).calls[0].with_overrides(w, h, depth)
        scene = gmlrt.DeviceScene(call.blob, 0)
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        sts = [scene.render_into(call.params, out.data_ptr()) for _ in range(12)]
        torch.cuda.synchronize()
        res[name] = {"first": sts[0].kernel_ms, "warm": [s.kernel_ms for s in sts[6:]], "way": sts[-1].handout_way,
                     "rays": [sts[-1].rays_primary, sts[-1].rays_shadow, sts[-1].rays_reflect, sts[-1].hits],
                     "sha": hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()}
        scene.close()
    print(json.dumps(res))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--worker":
        return worker(args[1].split(","))
    rounds, names = 2, list(CONFIGS)
    while args and args[0].startswith("--"):
        if args[0] == "--rounds": rounds = int(args[1])
        elif args[0] == "--scenes": names = args[1].split(",")
        args = args[2:]
    builds = [a.split("=", 1) for a in args]
    runs = {b: [] for b, _ in builds}
    for _ in range(rounds):
        for b, path in builds:
            env = dict(os.environ, GMLRT_LIB=os.path.abspath(path))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", ",".join(names)], env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(b, "FAILED", p.stderr[-400:])
                sys.exit(1)
            runs[b].append(json.loads(p.stdout.strip().splitlines()[-1]))
    bad = 0
    for name in names:
        ref = runs[builds[0][0]][0][name]
        line = "%-13s" % name
        for b, _ in builds:
            warm = sorted(x for r in runs[b] for x in r[name]["warm"])
            first = min(r[name]["first"] for r in runs[b])
            same = all(r[name]["sha"] == ref["sha"] and r[name]["rays"] == ref["rays"] for r in runs[b])
            bad += 0 if same else 1
            line += " | %s: best %.3f median %.3f first %.3f way %d%s" % (b, warm[0], warm[len(warm) // 2], first, runs[b][-1][name]["way"], "" if same else " DIFFERENT")
        print(line, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
