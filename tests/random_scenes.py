"""Seeded random GML scenes for parity runs: a handful of primitives under random transform chains, joined by random CSG
rules into a random tree (plus a floor), one to three lights of random kinds, a constant or a face/u/v-dependent surface."""
import random


def scene_text(seed, width=64, height=48, depth=2, lattice=False):
    """lattice: positions, sizes and angles come from a coarse grid (halves, right angles), so that faces of
    different primitives coincide exactly, rays run along axes and spans of zero length appear: the cases in which
    the order of equal distances and the 1e-10 rule of the CSG sweep decide the picture.  lattice=2: the same with
    rotations by eighths of a turn too."""
    rng = random.Random(seed)
    if lattice:
        f = lambda lo, hi: "%.1f" % (round(rng.uniform(lo, hi) * 2.0) / 2.0 or (0.5 if lo > 0 else 0.0))
    else:
        f = lambda lo, hi: "%.3f" % rng.uniform(lo, hi)
    out = []
    surfaces = ["{ /v /u /face %s %s %s point %s %s %s } /s%d" % (f(0.1, 1), f(0.1, 1), f(0.1, 1), f(0.3, 1), f(0.0, 0.6), f(1, 8), i)
                for i in range(2)]
    surfaces.append("{ /v /u /face u frac v frac face real 5.0 divf point 0.7 %s 3.0 } /s2" % f(0.0, 0.5))
    surfaces.append("{ /v /u /face u 3.0 mulf floor v 3.0 mulf floor addi 2 modi 0 eqi { 0.9 0.9 0.2 point } { 0.1 0.2 0.8 point } if "
                    "0.8 0.4 2.0 } /s3")
    out += surfaces

    def primitive():
        kind = rng.choice(["sphere", "cube", "cylinder", "cone", "sphere", "cylinder"])
        parts = ["s%d %s" % (rng.randrange(4), kind)]
        for _ in range(rng.randrange(1, 4)):
            op = rng.choice(["scale", "uscale", "rotatex", "rotatey", "rotatez", "translate"])
            if op == "scale":
                parts.append("%s %s %s scale" % (f(0.4, 1.6), f(0.4, 1.6), f(0.4, 1.6)))
            elif op == "uscale":
                parts.append("%s uscale" % f(0.5, 1.5))
            elif op == "translate":
                parts.append("%s %s %s translate" % (f(-0.6, 0.6), f(-0.6, 0.6), f(-0.6, 0.6)))
            else:
                angles = ["0.0", "90.0", "180.0", "-90.0"]
                if lattice == 2:  # eighths of a turn as well: rays along a cone's mantle, along the diagonals of cubes
                    angles += ["45.0", "-45.0", "135.0"]
                parts.append("%s %s" % (rng.choice(angles + ([] if lattice else [f(-180, 180)])), op))
        parts.append("%s %s %s translate" % (f(-1.5, 1.5), f(-1.0, 1.0), f(2.5, 5.0)))
        return " ".join(parts)

    def tree(n):
        if n == 1:
            return primitive()
        k = rng.randrange(1, n)
        return "%s %s %s" % (tree(k), tree(n - k), rng.choice(["union", "union", "intersect", "difference"]))

    out.append(tree(rng.randrange(2, 9)))
    out.append("s%d plane 0.0 %s 0.0 translate union /scene" % (rng.randrange(4), f(-2.0, -1.0)))
    lights = []
    for i in range(rng.randrange(1, 4)):
        kind = rng.choice(["light", "pointlight", "spotlight"])
        col = "%s %s %s point" % (f(0.3, 1), f(0.3, 1), f(0.3, 1))
        if kind == "light":
            out.append("%s %s %s point %s light /l%d" % (f(-1, 1), f(-1, -0.2), f(-0.5, 1), col, i))
        elif kind == "pointlight":
            out.append("%s %s %s point %s pointlight /l%d" % (f(-3, 3), f(1, 4), f(-1, 3), col, i))
        else:
            out.append("%s %s %s point %s %s %s point %s %s %s spotlight /l%d" % (f(-3, 3), f(2, 4), f(0, 3), f(-1, 1), f(-1, 0), f(3, 4.5),
                                                                              col, f(15, 50), f(1, 6), i))
        lights.append("l%d" % i)
    out.append('%s %s %s point [ %s ] scene %d %s %d %d "r.ppm" render' % (f(0.1, 0.5), f(0.1, 0.5), f(0.1, 0.5), " ".join(lights), depth,
                                                                          rng.choice(["60.0", "90.0", f(40, 110)]), width, height))
    return "\n".join(out)
