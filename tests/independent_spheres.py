"""An independent restatement of the reference's render path for ONE scene, in plain Python floats (IEEE f64).

Test infrastructure.  It shares nothing with the C++ oracle or the product's GML front-end: the scene of
/root/reference/gml/spheres.gml is written out by hand below (two unit spheres and a plane, translated and
united; one directional light), and the algorithm follows the reference's sources directly:
  camera, pixel loop      raytracer.rs:121-147
  trace, ambient/specular raytracer.rs:30-111
  Operator::intersect     primitives.rs:177-187, 253-315
  Sphere, Plane           primitives.rs:349-420, 945-1006
  Transform               transform.rs:9-36, 64-79
  DirectionalLight        lights.rs:27-50
  quantiser               raytracer.rs:21-25
  vecmath crate 0.3.1     dot products left to right, normalize = scale by 1/sqrt(dot)
A second opinion for the oracle (tests/test_oracle_kats.py compares the two byte for byte at a small size): two
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

    def translate(self, tx, ty, tz):
        tr = [[1.0, 0.0, 0.0, tx], [0.0, 1.0, 0.0, ty], [0.0, 0.0, 1.0, tz], [0.0, 0.0, 0.0, 1.0]]
        inv = [[1.0, 0.0, 0.0, -tx], [0.0, 1.0, 0.0, -ty], [0.0, 0.0, 1.0, -tz], [0.0, 0.0, 0.0, 1.0]]
        self.m = mat_mul(tr, self.m)
        self.inv_m = mat_mul(self.inv_m, inv)
        return self

    def intersect(self, raypos, raydir):
        d = mat_vec(self.inv_m, raydir, 0.0)
        scale = 1.0 / math.sqrt(dot(d, d))
        d = mul(d, scale)
        p = mat_vec(self.inv_m, raypos, 1.0)
        hits = []

        def hit(t, kind):
            # Intersection::new: distance = scale * odistance (primitives.rs:134-150)
            hits.append({"dist": scale * t, "t": t, "p": p, "d": d, "leaf": self, "type": kind, "orig": kind})

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

    def shade_inputs(self, opos):
        x, y, z = opos
        if self.kind == "sphere":
            u, v = math.atan2(x, z), (y + 1.0) / 2.0
            n_obj = opos
        else:
            u, v = x, z
            n_obj = (0.0, 1.0, 0.0)
        inv_t = [[self.inv_m[j][i] for j in range(4)] for i in range(4)]  # transform_normal: transpose(inv_m)
        return self.surface(0, u, v), normalize(mat_vec(inv_t, n_obj, 0.0))


class Union:
    def __init__(self, a, b):
        self.a, self.b = a, b

    def intersect(self, raypos, raydir):
        events = [(h, 1) for h in self.a.intersect(raypos, raydir)] + [(h, 2) for h in self.b.intersect(raypos, raydir)]
        events.sort(key=lambda e: e[0]["dist"])  # stable: ties keep obj1 before obj2
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
            now = in1 > 0 or in2 > 0
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


def build_scene():
    sphere_surface = lambda face, u, v: ((0.8, 0.2, v), 1.0, 0.2, 1.0)  # spheres.gml:11-14
    plane_surface = lambda face, u, v: ((1.0, 1.0, 1.0), 1.0, 0.0, 1.0)  # white matte (surface.ins:15-19, colors.ins)
    s1 = Leaf("sphere", sphere_surface, 1).translate(-1.2, 0.0, 3.0)
    s2 = Leaf("sphere", sphere_surface, 2).translate(1.2, 1.0, 3.0)
    p = Leaf("plane", plane_surface, 3).translate(0.0, -3.0, 0.0)
    scene = Union(s1, Union(s2, p))  # `s1 s2 p union union`
    light_dir = normalize(neg((1.0, -1.0, 0.0)))
    light_col = (1.0, 1.0, 1.0)
    return scene, light_dir, light_col


COUNTS = {"primary": 0, "shadow": 0, "reflect": 0}


def trace(scene, light, amb, depth, raypos, raydir):
    hits = scene.intersect(raypos, raydir)
    if not hits:
        return (0.0, 0.0, 0.0)
    h = hits[0]
    if h["type"] == EXIT:
        return (0.0, 0.0, 0.0)
    leaf = h["leaf"]
    opos = add(h["p"], mul(h["d"], h["t"]))
    (sc, kd, ks, n), normal = leaf.shade_inputs(opos)
    c = mul(cmul(amb, sc), kd)
    pos = mat_vec(leaf.m, opos, 1.0)
    if h["type"] != h["orig"]:
        normal = neg(normal)
    diffuse = specular = (0.0, 0.0, 0.0)
    ldir, lcol = light
    df = dot(normal, ldir)
    if df > 0.0:
        COUNTS["shadow"] += 1
        blockers = scene.intersect(add(pos, mul(ldir, 1e-7)), ldir)
        if not blockers:  # a directional light has no distance: any hit blocks it
            ic = cmul(sc, lcol)
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
        rc = trace(scene, light, amb, depth - 1, add(pos, mul(r, 1e-7)), r)
        return add(c, mul(cmul(rc, sc), ks))
    return c


def quantise(x):
    v = 255.0 * min(max(x, 0.0), 1.0)
    return 0 if v != v else int(v)


def render(width, height, depth=3, fov=90.0, amb=(0.4, 0.4, 0.4)):
    """Rows of (r, g, b) bytes, the unquantised pixels, and the ray counts."""
    scene, ldir, lcol = build_scene()
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
            px = trace(scene, (ldir, lcol), amb, depth, raypos, d)
            raw_row.append(px)
            row.append(tuple(quantise(v) for v in px))
        img.append(row)
        raw.append(raw_row)
    return img, raw, dict(COUNTS)
