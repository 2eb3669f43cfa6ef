"""The host's list of ray directions that are traced without bounds (gmlrt_scene.cpp: special_ray_classes) must
contain every ray that the reference's cylinder or cone treats by one of its special cases, whatever the
primitive's transform: such a ray can be given a hit outside the solid (primitives.rs:644-663, 778-792, 844-858),
which a bound would cull.  Rays are made in the primitive's object space inside each regime, taken to world space,
and must then be flagged by the kernel's own rule for the classes (restated below).  No GPU needed."""
import ctypes
import os
import struct

import numpy as np

LEAF_BYTES = 208
CYLINDER, CONE = 2, 3  # gmlrt_prim_type (include/gmlrt.h)
WORDS = 8              # doubles a class

SCENE = """
{ /v /u /face 0.8 0.8 0.8 point 0.5 0.5 2.0 } /s
s cylinder %s /a
s cone %s /b
s cylinder %s /c
s cone %s /d
a b union c union d union /scene
1.0 -1.0 0.5 point 1.0 1.0 1.0 point light /l
0.3 0.3 0.3 point [ l ] scene 2 90.0 32 32 "x.ppm" render
"""


def transforms(rng):
    ops = []
    for _ in range(rng.integers(1, 5)):
        k = rng.integers(0, 5)
        if k == 0:
            ops.append("%.4f %.4f %.4f scale" % tuple(rng.uniform(0.2, 3.0, 3)))
        elif k == 1:
            ops.append("%.4f uscale" % rng.uniform(0.05, 4.0))
        elif k == 2:
            ops.append("%.4f %.4f %.4f translate" % tuple(rng.uniform(-3, 3, 3)))
        else:
            ops.append("%.3f %s" % (rng.uniform(-180, 180), ["rotatex", "rotatey", "rotatez"][rng.integers(0, 3)]))
    return " ".join(ops)


def leaves_of(blob):
    n_leaves = struct.unpack_from("<I", blob, 16)[0]
    off_leaves = struct.unpack_from("<Q", blob, 48)[0]
    out = []
    for i in range(n_leaves):
        f = np.array(struct.unpack_from("<24d", blob, off_leaves + i * LEAF_BYTES))
        typ = struct.unpack_from("<I", blob, off_leaves + i * LEAF_BYTES + 192)[0]
        out.append((typ, f[:12].reshape(3, 4)[:, :3], f[12:].reshape(3, 4)[:, :3]))  # type, A (inverse), M
    return out


def classes_of(helpers, blob):
    helpers.lib.th_special_ray_classes.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                                   ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    buf = (ctypes.c_double * 4096)()
    na, nf = ctypes.c_uint(), ctypes.c_uint()
    n = helpers.lib.th_special_ray_classes(blob, buf, 4096, ctypes.byref(na), ctypes.byref(nf))
    c = np.array(buf[:n]).reshape(-1, WORDS)
    assert len(c) == na.value + nf.value
    return c[:na.value], c[na.value:]


def flagged(axes, forms, d):
    """The kernel's rule (gmlrt_kernels.hip: cylinder_special_case, cone_special_case) for unit directions d [n, 3]."""
    odd = np.zeros(len(d), dtype=bool)
    for a in axes:
        c = np.abs(d @ a[:3])
        odd |= (c >= a[3]) | (c <= a[4])
    for f in forms:
        q = (f[0] * d[:, 0] ** 2 + f[1] * d[:, 1] ** 2 + f[2] * d[:, 2] ** 2 + f[3] * d[:, 0] * d[:, 1] + f[4] * d[:, 0] * d[:, 2]
             + f[5] * d[:, 1] * d[:, 2])
        odd |= ~(np.abs(q) > f[6])
    return odd


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def test_every_ray_in_a_special_regime_is_flagged(gmlrt, helpers):
    rng = np.random.default_rng(7)
    n = 4000
    in_regime = 0
    for scene in range(60):
        text = SCENE % tuple(transforms(rng) for _ in range(4))
        blob = gmlrt.Program(text=text).calls[0].blob
        axes, forms = classes_of(helpers, blob)
        assert len(forms) >= 1 and len(axes) >= 1
        for typ, A, M in leaves_of(blob):
            if typ not in (CYLINDER, CONE):
                continue
            phi = rng.uniform(0, 2 * np.pi, n)
            ring = np.stack([np.cos(phi), np.zeros(n), np.sin(phi)], axis=1)
            cases = []
            # across the axis: |u_y| < 1e-7 of |u| (cylinder: primitives.rs:652; cone: a ray within the base plane)
            u = ring.copy()
            u[:, 1] = rng.uniform(-1, 1, n) * 0.999e-7
            cases.append(unit(u))
            if typ == CYLINDER:
                # along the axis: |u_y| + 1e-7 >= 1 (primitives.rs:644)
                s = np.sqrt(1.0 - (1.0 - 1e-7) ** 2) * rng.uniform(0, 0.999, n)
                u = ring * s[:, None]
                u[:, 1] = np.sqrt(1.0 - s * s) * rng.choice([-1.0, 1.0], n)
                cases.append(u)
            else:
                # along the mantle: |u_x^2 + u_z^2 - u_y^2| <= 1e-8 for a unit u (primitives.rs:778)
                a = rng.uniform(-1, 1, n) * 0.999e-8
                r = np.sqrt((1.0 + a) / 2.0)
                u = ring * r[:, None]
                u[:, 1] = np.sqrt((1.0 - a) / 2.0) * rng.choice([-1.0, 1.0], n)
                cases.append(u)
            for u in cases:
                d = unit(u @ M.T)  # the world direction of a ray whose object-space direction is u
                # what the reference sees: the inverse transform applied to d, normalised (primitives.rs:635-637)
                back = unit(d @ A.T)
                if typ == CYLINDER:
                    really = (np.abs(back[:, 1]) + 1e-7 >= 1.0) | (np.abs(back[:, 1]) < 1e-7)
                else:
                    really = (np.abs(back[:, 0] ** 2 + back[:, 2] ** 2 - back[:, 1] ** 2) <= 1e-8) | (np.abs(back[:, 1]) < 1e-7)
                in_regime += int(really.sum())
                assert flagged(axes, forms, d)[really].all(), (scene, typ)
    assert in_regime > 500000  # (nearly every ray made above is in its regime after the round trip too)


def test_few_rays_are_flagged(gmlrt, helpers):
    # a flagged ray costs its wave a walk through the whole scene: the classes must stay narrow
    rng = np.random.default_rng(11)
    worst = 0.0
    for scene in range(20):
        text = SCENE % tuple(transforms(rng) for _ in range(4))
        blob = gmlrt.Program(text=text).calls[0].blob
        axes, forms = classes_of(helpers, blob)
        d = unit(rng.normal(size=(400000, 3)))
        worst = max(worst, float(flagged(axes, forms, d).mean()))
    assert worst < 2e-4, worst


def test_classes_of_the_chess_scene(gmlrt, helpers):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = gmlrt.Program(path=os.path.join(root, "scenes", "chess.gml")).calls[0].blob
    axes, forms = classes_of(helpers, blob)
    # the pieces stand upright on the tilted board, a few parts lie across: two axis directions (along each and across
    # it) for the cylinders, the cones' base planes with them; a few cone shapes
    assert len(axes) <= 3 and 1 <= len(forms) <= 8
    d = unit(np.random.default_rng(3).normal(size=(4000000, 3)))
    assert flagged(axes, forms, d).mean() < 2e-6


def test_hits_of_unflagged_rays_stay_inside_the_widened_boxes(oracle):
    # The other half of the argument: a ray that is NOT flagged may still be near a regime, and the reference's answer for
    # it is then off by rounding (the cone's small root cancels like 1 / a; the cylinder's general case takes over right
    # at 1e-7).  Every hit the oracle's own arithmetic reports for such rays on the unit cone and the unit cylinder must lie
    # inside the object-space boxes the bounds are made from (gmlrt_scene.cpp: leaf_bound: a thousandth wider than the
    # primitive), for origins up to 1000 object units away.
    rng = np.random.default_rng(5)
    zero, one = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    worst = 0.0
    hits = 0
    for k in range(12000):
        phi = rng.uniform(0, 2 * np.pi)
        sgn = rng.choice([-1.0, 1.0])
        kind = k % 4
        if kind == 0:    # cone, just outside the band: 1e-8 < |a| <= 1e-5
            a = 10.0 ** rng.uniform(-8, -5) * 1.001 * rng.choice([-1.0, 1.0])
            r, y = np.sqrt((1.0 + a) / 2.0), np.sqrt((1.0 - a) / 2.0) * sgn
            typ = CONE
        elif kind == 1:  # cone, nearly within the base plane: 1e-7 <= |u_y| <= 1e-4
            y = 10.0 ** rng.uniform(-7, -4) * 1.001 * sgn
            r = np.sqrt(1.0 - y * y)
            typ = CONE
        elif kind == 2:  # cylinder, nearly across the axis
            y = 10.0 ** rng.uniform(-7, -4) * 1.001 * sgn
            r = np.sqrt(1.0 - y * y)
            typ = CYLINDER
        else:            # cylinder, nearly along the axis: 1 - |u_y| a little above 1e-7
            y = (1.0 - 1e-7 * 10.0 ** rng.uniform(0.001, 3)) * sgn
            r = np.sqrt(1.0 - y * y)
            typ = CYLINDER
        d = np.array([r * np.cos(phi), y, r * np.sin(phi)])
        d /= np.linalg.norm(d)
        # an origin from which the ray passes near the primitive: a point in its box, moved back along the ray
        through = np.array([rng.uniform(-1, 1), rng.uniform(0, 1), rng.uniform(-1, 1)])
        o = through - d * 10.0 ** rng.uniform(-1, 3)
        for is_exit, dist, face in oracle.kat_intersect(typ, zero, one, tuple(o), tuple(d)):
            p = o + dist * d
            out = max(abs(p[0]) - 1.0, abs(p[2]) - 1.0, -p[1], p[1] - 1.0)
            worst = max(worst, out)
            hits += 1
    assert hits > 5000
    assert worst < 5e-4, worst
