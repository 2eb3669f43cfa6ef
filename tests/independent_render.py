"""An independent restatement of the reference's render path in plain Python floats (IEEE f64).

Test infrastructure.  It shares nothing with the C++ oracle or the product's GML front-end: scenes are built by
hand with the calls below (build_spheres is /root/reference/gml/spheres.gml written out; build_mixed uses every
primitive, transform builder, CSG rule and light kind), and the algorithm follows the reference's sources directly:
  camera, pixel loop      raytracer.rs:121-147
  trace, ambient/specular raytracer.rs:30-111
  Operator::intersect     primitives.rs:177-187, 253-315
  Sphere, Cube, Cylinder, Cone, Plane   primitives.rs:349-420, 430-565, 587-743, 765-913, 945-1006
  Transform               transform.rs:9-36, 43-136
  lights                  lights.rs:27-120
  quantiser               raytracer.rs:21-25
  vecmath crate 0.3.1     dot products left to right, normalize = scale by 1/sqrt(dot)
A second opinion for the oracle (tests/test_oracle_kats.py compares the two bit for bit at small sizes): two
restatements written separately agreeing on every pixel is the strongest pin available without the Rust binary.
"""
import math

ENTRY, EXIT = 0, 1


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def mul(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def cmul(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def neg(a):
    return (-a[0], -a[1], -a[2])


def normalize(a):
    return mul(a, 1.0 / math.sqrt(dot(a, a)))


def mat_vec(m, v, w):
    # row_mat4_transform on [x, y, z, w], first three rows (transform.rs:9-19)
    return tuple(r[0] * v[0] + r[1] * v[1] + r[2] * v[2] + r[3] * w for r in m[:3])


def mat_mul(a, b):
    return [[a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j] for j in range(4)] for i in range(4)]


def identity():
    return [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]


class Leaf:
    """A primitive with its transform pair; translate() as transform.rs:64-79."""

    def __init__(self, kind, surface, ident):
        self.kind, self.surface, self.ident = kind, surface, ident
        self.m, self.inv_m = identity(), identity()

    def _apply(self, fwd, inv):
        self.m = mat_mul(fwd, self.m)
        self.inv_m = mat_mul(self.inv_m, inv)
        return self

    def translate(self, tx, ty, tz):
        return self._apply([[1.0, 0.0, 0.0, tx], [0.0, 1.0, 0.0, ty], [0.0, 0.0, 1.0, tz], [0.0, 0.0, 0.0, 1.0]],
                           [[1.0, 0.0, 0.0, -tx], [0.0, 1.0, 0.0, -ty], [0.0, 0.0, 1.0, -tz], [0.0, 0.0, 0.0, 1.0]])

    def scale(self, sx, sy, sz):
        return self._apply([[sx, 0.0, 0.0, 0.0], [0.0, sy, 0.0, 0.0], [0.0, 0.0, sz, 0.0], [0.0, 0.0, 0.0, 1.0]],
                           [[1.0 / sx, 0.0, 0.0, 0.0], [0.0, 1.0 / sy, 0.0, 0.0], [0.0, 0.0, 1.0 / sz, 0.0], [0.0, 0.0, 0.0, 1.0]])

    def uscale(self, s):
        return self.scale(s, s, s)

    def rotatex(self, deg):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        return self._apply([[1.0, 0.0, 0.0, 0.0], [0.0, c, -s, 0.0], [0.0, s, c, 0.0], [0.0, 0.0, 0.0, 1.0]],
                           [[1.0, 0.0, 0.0, 0.0], [0.0, c, s, 0.0], [0.0, -s, c, 0.0], [0.0, 0.0, 0.0, 1.0]])

    def rotatey(self, deg):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        return self._apply([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]],
                           [[c, 0.0, -s, 0.0], [0.0, 1.0, 0.0, 0.0], [s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]])

    def rotatez(self, deg):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        return self._apply([[c, -s, 0.0, 0.0], [s, c, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
                           [[c, s, 0.0, 0.0], [-s, c, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])

    def intersect(self, raypos, raydir):
        d = mat_vec(self.inv_m, raydir, 0.0)
        scale = 1.0 / math.sqrt(dot(d, d))
        d = mul(d, scale)
        p = mat_vec(self.inv_m, raypos, 1.0)
        hits = []

        def hit(t, kind, face=0):
            # Intersection::new: distance = scale * odistance (primitives.rs:134-150)
            hits.append({"dist": scale * t, "t": t, "p": p, "d": d, "leaf": self, "type": kind, "orig": kind, "face": face})

        def entry_exit(lo, hi):
            # (t, face) pairs: Entry at the first if t > 0, Exit at the second if t > 0
            if lo[0] > 0.0:
                hit(lo[0], ENTRY, lo[1])
            if hi[0] > 0.0:
                hit(hi[0], EXIT, hi[1])

        def quadratic(a, b, c):
            sq = b * b - 4.0 * a * c
            if sq < 0.0:
                return None
            root = math.sqrt(sq)
            t1, t2 = ((-b - root) / (2.0 * a), 0), ((-b + root) / (2.0 * a), 0)
            return (t2, t1) if t1 > t2 else (t1, t2)

        if self.kind == "cube":
            tmin = tmax = None
            e3 = sub((0.5, 0.5, 0.5), p)
            for i, (face1, face2) in enumerate(((3, 2), (4, 5), (1, 0))):
                e, f = e3[i], d[i]
                if abs(f) > 1e-15:
                    t1, t2 = (e + 0.5) / f, (e - 0.5) / f
                    if t1 > t2:
                        t1, t2, face1, face2 = t2, t1, face2, face1
                    if not (tmin is not None and tmin[0] >= t1):
                        tmin = (t1, face1)
                    if not (tmax is not None and tmax[0] <= t2):
                        tmax = (t2, face2)
                    if tmin[0] > tmax[0] or tmax[0] < 0.0:
                        return hits
                elif -e - 0.5 > 0.0 or -e + 0.5 < 0.0:
                    return hits
            if tmin is not None and tmin[0] > 0.0:
                hit(tmin[0], ENTRY, tmin[1])
            if tmax is not None and tmax[0] > 0.0:
                hit(tmax[0], EXIT, tmax[1])
            return hits
        if self.kind == "cylinder":
            (px, py, pz), (dx, dy, dz) = p, d

            def caps():
                dinv = 1.0 / dy
                t1, t2 = (-py * dinv, 2), ((-py + 1.0) * dinv, 1)
                return (t2, t1) if t1[0] > t2[0] else (t1, t2)

            def side():
                return quadratic(dx * dx + dz * dz, 2.0 * (px * dx + pz * dz), px * px + pz * pz - 1.0)

            if abs(dy) + 1e-7 >= 1.0:
                if 1.0 - px * px - pz * pz < 0.0:
                    return hits
                lo, hi = caps()
            elif abs(dy) < 1e-7:
                if py < 0.0 or py > 1.0:
                    return hits
                roots = side()
                if roots is None:
                    return hits
                lo, hi = roots
            else:
                roots = side()
                if roots is None:
                    return hits
                lo, hi = caps()
                if roots[0] > lo:  # tuple order: t first, then face
                    lo = roots[0]
                if roots[1] < hi:
                    hi = roots[1]
                if lo[0] > hi[0] or hi[0] < 0.0:
                    return hits
            entry_exit(lo, hi)
            return hits
        if self.kind == "cone":
            (px, py, pz), (dx, dy, dz) = p, d
            roots = quadratic(dx * dx + dz * dz - dy * dy, 2.0 * (px * dx + pz * dz - py * dy), px * px + pz * pz - py * py)
            if roots is None:
                return hits
            kept = [r for r in roots if 0.0 <= py + r[0] * dy <= 1.0]
            if not kept:
                return hits
            if len(kept) == 1:
                kept.append(((-py + 1.0) / dy, 1))  # the other crossing is the base
                if not (kept[1][0] >= kept[0][0]):
                    kept.reverse()
            entry_exit(kept[0], kept[1])
            return hits

        if self.kind == "sphere":
            s = dot(neg(p), d)
            lsq = dot(p, p)
            if s < 0.0 and lsq > 1.0:
                return hits
            msq = lsq - s * s
            if msq > 1.0:
                return hits
            q = math.sqrt(1.0 - msq)
            t1, t2 = s + q, s - q
            if t1 > t2:
                t1, t2 = t2, t1
            if t1 > 0.0:
                hit(t1, ENTRY)
            if t2 > 0.0:
                hit(t2, EXIT)
        else:  # plane y <= 0
            denom = dot((0.0, 1.0, 0.0), d)
            if abs(denom) < 1e-7:
                return hits
            t = -dot((0.0, 1.0, 0.0), p) / denom
            if t < 0.0:
                return hits
            hit(t, EXIT if denom > 0.0 else ENTRY)
        return hits

    def shade_inputs(self, opos, face):
        x, y, z = opos
        if self.kind == "sphere":
            u, v, n_obj = math.atan2(x, z), (y + 1.0) / 2.0, opos
        elif self.kind == "cube":
            u, v = ((x, y), (x, y), (z, y), (z, y), (x, z), (x, z))[face]
            n_obj = ((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (-1.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, -1.0, 0.0))[face]
        elif self.kind in ("cylinder", "cone"):
            if face == 0:
                u, v = math.atan2(x, z), y
                n_obj = (x, 0.0, z) if self.kind == "cylinder" else (x, -y, z)
            else:
                u, v = (x + 1.0) / 2.0, (z + 1.0) / 2.0
                n_obj = (0.0, -1.0, 0.0) if (self.kind == "cylinder" and face == 2) else (0.0, 1.0, 0.0)
        else:
            u, v, n_obj = x, z, (0.0, 1.0, 0.0)
        inv_t = [[self.inv_m[j][i] for j in range(4)] for i in range(4)]  # transform_normal: transpose(inv_m)
        return self.surface(face, u, v), normalize(mat_vec(inv_t, n_obj, 0.0))


class Operator:
    RULES = {"union": lambda a, b: a or b, "intersect": lambda a, b: a and b, "difference": lambda a, b: a and not b}

    def __init__(self, rule, a, b):
        self.rule, self.a, self.b = self.RULES[rule], a, b

    # transforms are forwarded to both operands (primitives.rs:213-238)
    def _both(self, name, *args):
        getattr(self.a, name)(*args)
        getattr(self.b, name)(*args)
        return self

    def translate(self, *a): return self._both("translate", *a)
    def scale(self, *a): return self._both("scale", *a)
    def uscale(self, *a): return self._both("uscale", *a)
    def rotatex(self, *a): return self._both("rotatex", *a)
    def rotatey(self, *a): return self._both("rotatey", *a)
    def rotatez(self, *a): return self._both("rotatez", *a)

    def intersect(self, raypos, raydir):
        events = [(h, 1) for h in self.a.intersect(raypos, raydir)] + [(h, 2) for h in self.b.intersect(raypos, raydir)]
        events.sort(key=lambda e: e[0]["dist"])  # stable: ties keep obj1 before obj2
        if len(events) > MAX_EVENTS_SEEN[0]:
            MAX_EVENTS_SEEN[0] = len(events)
        in1 = in2 = 0
        inside = False
        prevt = 0.0
        out = []
        for h, who in events:
            step = 1 if h["type"] == ENTRY else -1
            if who == 1:
                in1 += step
            else:
                in2 += step
            now = self.rule(in1 > 0, in2 > 0)
            if inside and not now:
                if h["dist"] - prevt < 1e-10:
                    out.pop()
                else:
                    out.append(dict(h, type=EXIT))
            if not inside and now:
                out.append(dict(h, type=ENTRY))
                prevt = h["dist"]
            inside = now
        return out


class Directional:
    def __init__(self, d, color):
        self.dir, self.color = normalize(neg(d)), color

    def direction(self, pos):
        return self.dir, None

    def intensity(self, pos):
        return self.color


class Point:
    def __init__(self, p, color):
        self.p, self.color = p, color

    def direction(self, pos):
        d = sub(self.p, pos)
        dl = math.sqrt(dot(d, d))
        return mul(d, 1.0 / dl), dl

    def intensity(self, pos):
        d = sub(self.p, pos)
        return mul(self.color, 100.0 / (99.0 + dot(d, d)))


class Spot(Point):
    def __init__(self, p, at, color, cutoff_deg, exp):
        Point.__init__(self, p, color)
        self.axis, self.coscut, self.exp = normalize(sub(at, p)), math.cos(math.radians(cutoff_deg)), exp

    def intensity(self, pos):
        d = sub(self.p, pos)
        dsq = dot(d, d)
        inv = 1.0 / math.sqrt(dsq)
        if -dot(d, self.axis) * inv < self.coscut:
            return (0.0, 0.0, 0.0)
        return mul(self.color, math.pow(dot(self.axis, mul(d, -inv)), self.exp) * 100.0 / (99.0 + dsq))


def build_spheres():
    """gml/spheres.gml: 0.4 ambient, depth 3, fov 90."""
    sphere_surface = lambda face, u, v: ((0.8, 0.2, v), 1.0, 0.2, 1.0)  # spheres.gml:11-14
    plane_surface = lambda face, u, v: ((1.0, 1.0, 1.0), 1.0, 0.0, 1.0)  # white matte (surface.ins:15-19, colors.ins)
    s1 = Leaf("sphere", sphere_surface, 1).translate(-1.2, 0.0, 3.0)
    s2 = Leaf("sphere", sphere_surface, 2).translate(1.2, 1.0, 3.0)
    p = Leaf("plane", plane_surface, 3).translate(0.0, -3.0, 0.0)
    scene = Operator("union", s1, Operator("union", s2, p))  # `s1 s2 p union union`
    return scene, [Directional((1.0, -1.0, 0.0), (1.0, 1.0, 1.0))]


def frac(x):
    return x - math.trunc(x)


MIXED_GML = """
{ /v /u /face u frac v frac face real 5.0 divf point 0.8 0.3 4.0 } /m
m cube 1.5 1.0 1.2 scale 30.0 rotatey -1.8 -1.0 3.5 translate
m sphere 0.7 uscale -1.0 0.0 3.6 translate difference
m cylinder 0.5 1.6 0.5 scale 20.0 rotatez 1.4 -1.0 4.0 translate union
m cone 0.0 -1.0 0.0 translate 180.0 rotatex 0.8 1.4 0.8 scale 0.2 -1.0 2.6 translate
m plane 0.0 0.1 0.0 translate intersect union
m plane 0.0 -1.0 0.0 translate union /scene
1.0 -1.0 0.5 point 0.5 0.5 0.5 point light /l1
-2.0 2.0 1.0 point 0.9 0.8 0.7 point pointlight /l2
2.0 3.0 2.0 point 0.5 -1.0 3.5 point 1.0 1.0 0.6 point 25.0 3.0 spotlight /l3
0.2 0.25 0.3 point [ l1 l2 l3 ] scene 2 80.0 %d %d "mixed.ppm" render
"""


def build_mixed():
    """The scene of MIXED_GML, built by hand: every primitive, builder, rule and light kind."""
    m = lambda face, u, v: ((frac(u), frac(v), float(face) / 5.0), 0.8, 0.3, 4.0)
    cube = Leaf("cube", m, 1).scale(1.5, 1.0, 1.2).rotatey(30.0).translate(-1.8, -1.0, 3.5)
    bite = Leaf("sphere", m, 2).uscale(0.7).translate(-1.0, 0.0, 3.6)
    cyl = Leaf("cylinder", m, 3).scale(0.5, 1.6, 0.5).rotatez(20.0).translate(1.4, -1.0, 4.0)
    cone = Leaf("cone", m, 4).translate(0.0, -1.0, 0.0).rotatex(180.0).scale(0.8, 1.4, 0.8).translate(0.2, -1.0, 2.6)
    cut = Leaf("plane", m, 5).translate(0.0, 0.1, 0.0)
    floor = Leaf("plane", m, 6).translate(0.0, -1.0, 0.0)
    scene = Operator("union", Operator("union", Operator("union", Operator("difference", cube, bite), cyl),
                                       Operator("intersect", cone, cut)), floor)
    # the evaluator hands `render` the light array reversed (evaluator.rs:531-542): sums over lights run l3, l2, l1
    lights = [Spot((2.0, 3.0, 2.0), (0.5, -1.0, 3.5), (1.0, 1.0, 0.6), 25.0, 3.0),
              Point((-2.0, 2.0, 1.0), (0.9, 0.8, 0.7)), Directional((1.0, -1.0, 0.5), (0.5, 0.5, 0.5))]
    return scene, lights


# ---- a chess piece with its procedural surface, and the board, built by hand (gml/chess.gml:17-60, 76-104, 225-271)
def gml_floor(x):
    return int(math.floor(x))  # `f.floor() as i64` (evaluator.rs:388); the arguments here are far from the i64 range


def gml_modi(a, b):
    q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)  # Rust's `/` truncates towards zero (evaluator.rs:216-218)
    return a - b * q


def gml_asin_deg(x):
    return math.asin(x) * (180.0 / math.pi)  # evaluator.rs:335-339: degrees


def figure_surface(col1, col2):
    """figureSurface of gml/chess.gml:28-43 written out in Python instead of being evaluated as GML."""
    def surface(face, u, v):
        if face == 0:  # rotational: stripes according to the angle u (radians: primitives.rs:403-412, 726-734)
            col = col1 if gml_modi(gml_floor(u * 12.0), 2) == 1 else col2
        else:          # flat: rays according to the angle from the centre to (u, v)
            uu, vv = u - 0.5, v - 0.5
            b = uu / math.sqrt(uu * uu + vv * vv)
            a = gml_asin_deg(b) if 0.0 < vv else 360.0 - gml_asin_deg(b)
            col = col1 if gml_modi(gml_floor((a + 180.0) / 30.0), 2) == 1 else col2
        return (col, 0.7, 0.3, 1.0)
    return surface


def board_surface(face, u, v):
    """gml/chess.gml:49-60."""
    black1, white2 = (0.2, 0.2, 0.3), (1.0, 1.0, 0.8)
    if 3 < face:
        lhs, k = 0, gml_floor(u * 8.0) + gml_floor(v * 8.0)
    else:
        lhs, k = (0 if gml_modi(face, 2) == 0 else 1), gml_floor(u * 8.0)
    return (black1 if lhs == gml_modi(k, 2) else white2, 0.4, 0.6, 0.5)


TOWER_GML = """
{ /y /x x x mulf y y mulf addf sqrt } /dist
0.2 0.2 0.3 point /black1  0.4 0.4 0.5 point /black2  0.7 0.7 0.5 point /white1  1.0 1.0 0.8 point /white2
{ /col2 /col1
  { /v /u /face
    face 0 eqi
    { u 12.0 mulf floor 2 modi 1 eqi { col1 } { col2 } if }
    { u 0.5 subf /u v 0.5 subf /v
      u u v dist apply divf /b
      0.0 v lessf { b asin } { 360.0 b asin subf } if 180.0 addf 30.0 divf
      floor 2 modi 1 eqi { col1 } { col2 } if }
    if
    0.7 0.3 1.0 }
} /figureSurface
{ /v /u /face
  3 face lessi
  { 0 u 8.0 mulf floor v 8.0 mulf floor addi }
  { face 2 modi 0 eqi { 0 } { 1 } if u 8.0 mulf floor } if 2 modi eqi { black1 } { white2 } if
  0.4 0.6 0.5
} cube -0.5 -1.0 -0.5 translate 8.0 0.3 8.0 scale /board
{ /col2 /col1
  col1 col2 figureSurface apply /surface
  col1 col1 figureSurface apply /surface1
  col2 col2 figureSurface apply /surface2
  surface cylinder 1.0 0.75 1.0 scale
  surface cone 0.0 -1.0 0.0 translate 180.0 rotatez 90.0 rotatey
  1.0 7.5 1.0 scale union
  surface cylinder 1.0 2.8 1.0 scale intersect
  surface cylinder 0.0 3.0 0.0 translate union
  surface cylinder 0.7 4.0 0.7 scale
  { /rot
    surface1 plane 90.0 rotatex
    surface2 plane -90.0 rotatex 30.0 rotatey intersect rot rotatey
  } /apex
  15.0 apex apply 75.0 apex apply union 135.0 apex apply union
  195.0 apex apply union 255.0 apex apply union 315.0 apex apply union
  surface cylinder intersect union
  0.0 3.5 0.0 translate difference
  0.4 uscale
} /tower
white1 white2 tower apply /whiteTower
black1 black2 tower apply /blackTower
board
whiteTower -1.5 0.0 -1.5 translate union
blackTower 1.5 0.0 0.5 translate union
30.0 rotatey -20.0 rotatex 0.4 uscale 0.3 0.0 3.0 translate /scene
0.0 0.0 -1.0 point 1.0 1.0 1.0 point pointlight /l
0.33 0.33 0.33 point [ l ] scene %d 60.0 %d %d "tower.ppm" render
"""


def build_towers():
    """The scene of TOWER_GML built by hand: the board and two towers of gml/chess.gml (a piece of nested
    intersections, unions and a difference whose operands share surfaces exactly: the base cylinder, the 2.8-high
    cylinder and the wall cone's base all lie on x^2 + z^2 = 1; the platform's bottom coincides with the wall's cut)."""
    black1, black2, white1, white2 = (0.2, 0.2, 0.3), (0.4, 0.4, 0.5), (0.7, 0.7, 0.5), (1.0, 1.0, 0.8)
    ids = iter(range(1, 1000))

    def tower(col1, col2):
        surface, surface1, surface2 = figure_surface(col1, col2), figure_surface(col1, col1), figure_surface(col2, col2)
        base = Leaf("cylinder", surface, next(ids)).scale(1.0, 0.75, 1.0)
        wall = Leaf("cone", surface, next(ids)).translate(0.0, -1.0, 0.0).rotatez(180.0).rotatey(90.0).scale(1.0, 7.5, 1.0)
        body = Operator("union", base, wall)
        body = Operator("intersect", body, Leaf("cylinder", surface, next(ids)).scale(1.0, 2.8, 1.0))
        body = Operator("union", body, Leaf("cylinder", surface, next(ids)).translate(0.0, 3.0, 0.0))
        bore = Leaf("cylinder", surface, next(ids)).scale(0.7, 4.0, 0.7)

        def apex(rot):
            a = Leaf("plane", surface1, next(ids)).rotatex(90.0)
            b = Leaf("plane", surface2, next(ids)).rotatex(-90.0).rotatey(30.0)
            return Operator("intersect", a, b).rotatey(rot)

        holes = apex(15.0)
        for rot in (75.0, 135.0, 195.0, 255.0, 315.0):
            holes = Operator("union", holes, apex(rot))
        holes = Operator("intersect", holes, Leaf("cylinder", surface, next(ids)))
        cut = Operator("union", bore, holes).translate(0.0, 3.5, 0.0)
        return Operator("difference", body, cut).uscale(0.4)

    board = Leaf("cube", board_surface, next(ids)).translate(-0.5, -1.0, -0.5).scale(8.0, 0.3, 8.0)
    scene = Operator("union", Operator("union", board, tower(white1, white2).translate(-1.5, 0.0, -1.5)),
                     tower(black1, black2).translate(1.5, 0.0, 0.5))
    scene.rotatey(30.0).rotatex(-20.0).uscale(0.4).translate(0.3, 0.0, 3.0)
    return scene, [Point((0.0, 0.0, -1.0), (1.0, 1.0, 1.0))]


# ---- more than twenty events with pairwise equal distances in one operator
TIES_GML = """
{ /v /u /face 0.9 0.2 0.1 point 0.8 0.3 2.0 } /red
{ /v /u /face 0.1 0.3 0.9 point 0.8 0.3 2.0 } /blue
{ /v /u /face 0.2 0.8 0.2 point 1.0 0.0 1.0 } /green
{ /s s sphere 0.45 uscale -2.5 0.0 0.0 translate
  s sphere 0.45 uscale -1.5 0.0 0.0 translate union  s sphere 0.45 uscale -0.5 0.0 0.0 translate union
  s sphere 0.45 uscale 0.5 0.0 0.0 translate union   s sphere 0.45 uscale 1.5 0.0 0.0 translate union
  s sphere 0.45 uscale 2.5 0.0 0.0 translate union } /row
red row apply blue row apply union
green cube -0.5 -0.5 -0.5 translate 5.5 0.6 2.0 scale intersect
90.0 rotatey 0.05 0.03 4.5 translate /scene
1.0 -1.0 1.0 point 1.0 1.0 1.0 point light /l
0.3 0.3 0.3 point [ l ] scene 2 60.0 %d %d "ties.ppm" render
"""


def build_ties():
    """The scene of TIES_GML: two rows of six spheres that coincide sphere for sphere (one row red, one blue), united,
    then cut by a slab.  A ray along the rows meets up to 24 events in the union node, every distance twice: which of
    two events at the same distance comes first decides whose surface is shaded.  This code, the oracle and the device
    all take the sort of primitives.rs:271 (`sort_unstable_by`) to keep the order obj1-before-obj2 on ties, which is
    what Rust's insertion sort does for at most 20 elements; for longer lists Rust leaves the order unspecified."""
    red = lambda face, u, v: ((0.9, 0.2, 0.1), 0.8, 0.3, 2.0)
    blue = lambda face, u, v: ((0.1, 0.3, 0.9), 0.8, 0.3, 2.0)
    green = lambda face, u, v: ((0.2, 0.8, 0.2), 1.0, 0.0, 1.0)
    ids = iter(range(1, 100))

    def row(surface):
        node = Leaf("sphere", surface, next(ids)).uscale(0.45).translate(-2.5, 0.0, 0.0)
        for x in (-1.5, -0.5, 0.5, 1.5, 2.5):
            node = Operator("union", node, Leaf("sphere", surface, next(ids)).uscale(0.45).translate(x, 0.0, 0.0))
        return node

    both = Operator("union", row(red), row(blue))
    slab = Leaf("cube", green, next(ids)).translate(-0.5, -0.5, -0.5).scale(5.5, 0.6, 2.0)
    scene = Operator("intersect", both, slab).rotatey(90.0).translate(0.05, 0.03, 4.5)
    return scene, [Directional((1.0, -1.0, 1.0), (1.0, 1.0, 1.0))]


MAX_EVENTS_SEEN = [0]

COUNTS = {"primary": 0, "shadow": 0, "reflect": 0}


def trace(scene, lights, amb, depth, raypos, raydir):
    hits = scene.intersect(raypos, raydir)
    if not hits:
        return (0.0, 0.0, 0.0)
    h = hits[0]
    if h["type"] == EXIT:
        return (0.0, 0.0, 0.0)
    leaf = h["leaf"]
    opos = add(h["p"], mul(h["d"], h["t"]))
    (sc, kd, ks, n), normal = leaf.shade_inputs(opos, h["face"])
    c = mul(cmul(amb, sc), kd)
    pos = mat_vec(leaf.m, opos, 1.0)
    if h["type"] != h["orig"]:
        normal = neg(normal)
    diffuse = specular = (0.0, 0.0, 0.0)
    for light in lights:
        ldir, ldist = light.direction(pos)
        df = dot(normal, ldir)
        if df > 0.0:
            COUNTS["shadow"] += 1
            blockers = scene.intersect(add(pos, mul(ldir, 1e-7)), ldir)
            if not blockers or (ldist is not None and ldist < blockers[0]["dist"]):
                ic = cmul(sc, light.intensity(pos))
                if kd > 0.0:
                    diffuse = add(diffuse, mul(ic, df))
                if ks > 0.0:
                    halfway = normalize(add(ldir, normalize(sub(raypos, pos))))
                    sp = dot(normal, halfway)
                    specular = add(specular, mul(ic, math.pow(sp, n)) if sp > 0.0 else (0.0, 0.0, 0.0))
    c = add(c, add(mul(diffuse, kd), mul(specular, ks)))
    if ks > 0.0 and depth > 0:
        COUNTS["reflect"] += 1
        r = normalize(sub(raydir, mul(normal, 2.0 * dot(raydir, normal))))
        rc = trace(scene, lights, amb, depth - 1, add(pos, mul(r, 1e-7)), r)
        return add(c, mul(cmul(rc, sc), ks))
    return c


def quantise(x):
    v = 255.0 * min(max(x, 0.0), 1.0)
    return 0 if v != v else int(v)


def render(build, width, height, depth, fov, amb):
    """Rows of (r, g, b) bytes, the unquantised pixels, and the ray counts."""
    scene, lights = build()
    for k in COUNTS:
        COUNTS[k] = 0
    raypos = (0.0, 0.0, -1.0)
    w_world = 2.0 * math.tan(0.5 * math.radians(fov))
    h_world = height * w_world / width
    c_x, c_y, pw = -0.5 * w_world, 0.5 * h_world, w_world / width
    img, raw = [], []
    for iy in range(height):
        row, raw_row = [], []
        for ix in range(width):
            d = normalize((c_x + (ix + 0.5) * pw, c_y - (iy + 0.5) * pw, 1.0))
            COUNTS["primary"] += 1
            px = trace(scene, lights, amb, depth, raypos, d)
            raw_row.append(px)
            row.append(tuple(quantise(v) for v in px))
        img.append(row)
        raw.append(raw_row)
    return img, raw, dict(COUNTS)
