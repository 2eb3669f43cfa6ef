"""Pins the CPU oracle: the reference's own unit-test facts for this path plus analytic closed forms.

Reference facts restated:
  primitives.rs:1016-1031  Intersection::new(scale=2, odistance=3).distance == 6
  primitives.rs:1033-1043  unit sphere translated to z=+5, ray (0,0,0)->(0,0,1): Entry 4, Exit 6
  raytracer.rs:230-233     normalize([7,0,0]) == [1,0,0]
  evaluator.rs:1043-1053   directional light (1,0,0): direction (-1,0,0), intensity = colour
  raytracer.rs:21-25       quantiser: truncating `as u8` of 255*clamp(c,0,1)
The reference holds no image-level fixture (its .gitignore drops *.ppm), so the images under tests/golden/ are
regression fixtures made by this oracle (tests/golden/make_golden.py), not reference output: image-level parity
with the Rust binary is UNPINNED.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES

SPHERE, CUBE, CYLINDER, CONE, PLANE = range(5)
ONE = (1.0, 1.0, 1.0)
ZERO = (0.0, 0.0, 0.0)


def test_sphere_translated_kat(oracle):
    hits = oracle.kat_intersect(SPHERE, (0, 0, 5), ONE, ZERO, (0, 0, 1))
    assert hits == [(False, 4.0, 0), (True, 6.0, 0)]


def test_distance_is_scale_times_object_distance(oracle):
    # a sphere scaled by 2: object-space t is half the world distance; distance = scale * odistance
    hits = oracle.kat_intersect(SPHERE, (0, 0, 10), (2.0, 2.0, 2.0), ZERO, (0, 0, 1))
    assert hits == [(False, 8.0, 0), (True, 12.0, 0)]


def test_origin_inside_gives_lone_exit(oracle):
    assert oracle.kat_intersect(SPHERE, ZERO, ONE, ZERO, (0, 0, 1)) == [(True, 1.0, 0)]
    assert oracle.kat_intersect(SPHERE, (0, 0, -5), ONE, ZERO, (0, 0, 1)) == []       # behind the ray


def test_cube_faces(oracle):
    # unit cube [0,1]^3 pushed to z in [2,3]; ray along +z through its middle enters face 0 (z=0), leaves face 1
    assert oracle.kat_intersect(CUBE, (-0.5, -0.5, 2.0), ONE, ZERO, (0, 0, 1)) == [(False, 2.0, 0), (True, 3.0, 1)]
    # along +x: faces 2 (x=0) and 3 (x=1); along -y from above: 4 (y=1) then 5 (y=0)
    assert oracle.kat_intersect(CUBE, (2.0, -0.5, -0.5), ONE, ZERO, (1, 0, 0)) == [(False, 2.0, 2), (True, 3.0, 3)]
    assert oracle.kat_intersect(CUBE, (-0.5, -3.0, -0.5), ONE, ZERO, (0, -1, 0)) == [(False, 2.0, 4), (True, 3.0, 5)]
    assert oracle.kat_intersect(CUBE, (5.0, 5.0, 2.0), ONE, ZERO, (0, 0, 1)) == []     # parallel slabs miss


def test_cylinder_regimes(oracle):
    # axis-parallel ray from above: top cap (face 1) then bottom cap (face 2)
    assert oracle.kat_intersect(CYLINDER, (0, -3.0, 0), ONE, ZERO, (0, -1, 0)) == [(False, 2.0, 1), (True, 3.0, 2)]
    # orthogonal ray at mid height: side in and out (face 0)
    assert oracle.kat_intersect(CYLINDER, (0, -0.5, 5.0), ONE, ZERO, (0, 0, 1)) == [(False, 4.0, 0), (True, 6.0, 0)]
    assert oracle.kat_intersect(CYLINDER, (0, 2.0, 5.0), ONE, ZERO, (0, 0, 1)) == []   # above the caps
    # general ray: in through the top cap, out through the side
    d = 1.0 / math.sqrt(2.0)
    (e0, t0, f0), (e1, t1, f1) = oracle.kat_intersect(CYLINDER, (0, -2.0, 0.5), ONE, ZERO, (0, -d, d))
    assert (e0, f0, e1, f1) == (False, 1, True, 0)
    assert t0 == pytest.approx(math.sqrt(2.0)) and t1 == pytest.approx(1.5 * math.sqrt(2.0))


def test_cone_mantle_and_base(oracle):
    # cone x^2+z^2=y^2, 0<=y<=1.  Horizontal ray at y=0.5: mantle at z=-0.5 and z=+0.5
    assert oracle.kat_intersect(CONE, (0, -0.5, 5.0), ONE, ZERO, (0, 0, 1)) == [(False, 4.5, 0), (True, 5.5, 0)]
    # vertical ray a quarter off the axis, from above: base cap (face 1) at y=1, then the mantle (face 0) at y=0.25
    hits = oracle.kat_intersect(CONE, (-0.25, -3.0, 0.0), ONE, ZERO, (0, -1, 0))
    assert hits == [(False, 2.0, 1), (True, 2.75, 0)]
    # down the axis itself the quadratic has a double root at the apex: Entry and Exit coincide, both on the mantle
    assert oracle.kat_intersect(CONE, (0.0, -3.0, 0.0), ONE, ZERO, (0, -1, 0)) == [(False, 3.0, 0), (True, 3.0, 0)]


def test_plane_entry_exit(oracle):
    # half-space y <= 0 moved to y <= -3: looking down we enter at 3; looking up from below we leave (Exit)
    assert oracle.kat_intersect(PLANE, (0, -3.0, 0), ONE, ZERO, (0, -1, 0)) == [(False, 3.0, 0)]
    assert oracle.kat_intersect(PLANE, (0, 3.0, 0), ONE, ZERO, (0, 1, 0)) == [(True, 3.0, 0)]
    assert oracle.kat_intersect(PLANE, (0, -3.0, 0), ONE, ZERO, (1, 0, 0)) == []       # |denom| < 1e-7


def test_quantiser(oracle):
    for v, b in [(1.0, 255), (0.999, 254), (0.5, 127), (-0.1, 0), (float("nan"), 0), (2.0, 255), (1.0 / 255.0, 1),
                 (0.0, 0)]:
        assert oracle.quantise(v) == b, v


def test_camera_and_background(oracle):
    # a scene whose only object is far off to the side: every pixel black, exactly w*h primary rays, no others
    text = ('{ /v /u /face 1.0 1.0 1.0 point 1.0 0.0 1.0 } sphere 100.0 0.0 0.0 translate /s '
            '1.0 1.0 1.0 point [ ] s 2 90.0 16 8 "x.ppm" render')
    img, f64, cnt = oracle.OracleProgram(text=text).render(0, want_f64=True)
    assert img.shape == (8, 16, 3) and not img.any() and not f64.any()
    assert cnt == {"primary": 128, "shadow": 0, "reflect": 0}


def test_ambient_only_sphere_is_exact(oracle):
    # fov 90, sphere of radius 1 at z=3, ambient (0.5,0.25,1) * colour (1,1,1) * kd 0.8, no lights, ks 0:
    # every hit pixel is exactly (0.4, 0.2, 0.8); the centre pixels hit, the corner misses
    text = ('{ /v /u /face 1.0 1.0 1.0 point 0.8 0.0 1.0 } sphere 0.0 0.0 3.0 translate /s '
            '0.5 0.25 1.0 point [ ] s 0 90.0 64 64 "x.ppm" render')
    img, f64, cnt = oracle.OracleProgram(text=text).render(0, want_f64=True)
    assert tuple(f64[32, 32]) == (0.4, 0.2, 0.8) and tuple(img[32, 32]) == (102, 51, 204)
    assert tuple(f64[0, 0]) == (0.0, 0.0, 0.0)
    hit = f64[..., 0] > 0
    # silhouette: the sphere subtends asin(1/4) from the eye at z=-1; pixel (x+.5)*pw - 1 in tan units
    yy, xx = np.mgrid[0:64, 0:64]
    dx, dy = -1.0 + (xx + 0.5) / 32.0, 1.0 - (yy + 0.5) / 32.0
    cosang = 1.0 / np.sqrt(dx * dx + dy * dy + 1.0)
    assert (hit == (cosang > math.cos(math.asin(0.25)))).all()
    assert cnt["reflect"] == 0 and cnt["shadow"] == 0


def test_directional_light_diffuse_closed_form(oracle):
    # light travelling along +z hits the sphere's front pole head-on: N.L = 1 at the centre pixel of an odd image
    text = ('0.0 0.0 1.0 point 1.0 1.0 1.0 point light /l '
            '{ /v /u /face 1.0 0.5 0.25 point 0.5 0.0 1.0 } sphere 0.0 0.0 3.0 translate /s '
            '0.0 0.0 0.0 point [ l ] s 0 90.0 33 33 "x.ppm" render')
    _, f64, cnt = oracle.OracleProgram(text=text).render(0, want_f64=True)
    assert f64[16, 16] == pytest.approx([0.5, 0.25, 0.125], abs=1e-15)
    assert cnt["shadow"] > 0


def test_lights_are_walked_in_reverse_source_order(oracle):
    # two coloured lights: swapping them in the array must not change the (commutative up to rounding) image much,
    # but the ray COUNTS are identical and shadow rays = hits facing each light
    a = ('0.0 0.0 1.0 point 1.0 0.0 0.0 point light /l1 1.0 0.0 1.0 point 0.0 0.0 1.0 point light /l2 '
         '{ /v /u /face 1.0 1.0 1.0 point 1.0 0.0 1.0 } sphere 0.0 0.0 3.0 translate /s '
         '0.0 0.0 0.0 point [ %s ] s 0 90.0 32 32 "x.ppm" render')
    _, fa, ca = oracle.OracleProgram(text=a % "l1 l2").render(0, want_f64=True)
    _, fb, cb = oracle.OracleProgram(text=a % "l2 l1").render(0, want_f64=True)
    assert ca == cb and np.abs(fa - fb).max() < 1e-15


def test_mirror_recursion_counts(oracle):
    # two facing mirrors (planes) and depth d: a ray bounces exactly d times
    text = ('{ /v /u /face 1.0 1.0 1.0 point 0.0 0.5 1.0 } /mir '
            'mir plane 0.0 -1.0 0.0 translate mir plane 180.0 rotatex 0.0 1.0 0.0 translate union /s '
            '1.0 1.0 1.0 point [ ] s %d 90.0 8 8 "x.ppm" render')
    for depth in (0, 1, 4):
        _, _, cnt = oracle.OracleProgram(text=text % depth).render(0)
        assert cnt["primary"] == 64 and cnt["reflect"] == 64 * depth


def test_csg_difference_and_sliver_rule(oracle):
    # difference of two identical cubes: every span has zero length and is popped (primitives.rs:295-298) -> black
    text = ('{ /v /u /face 1.0 1.0 1.0 point 1.0 0.0 1.0 } /m m cube m cube difference -0.5 -0.5 2.0 translate /s '
            '1.0 1.0 1.0 point [ ] s 0 90.0 16 16 "x.ppm" render')
    img, _, _ = oracle.OracleProgram(text=text).render(0)
    assert not img.any()
    # intersect of a cube with itself is the cube; union too
    for op in ("intersect", "union"):
        t2 = text.replace("difference", op)
        img2, _, _ = oracle.OracleProgram(text=t2).render(0)
        assert img2[8, 8].tolist() == [255, 255, 255] and not img2[0, 0].any()


@pytest.mark.parametrize("name", ["spheres", "spotlight", "house", "cone-fractal", "chess", "fractal", "large"])
def test_golden_regression(oracle, name):
    meta = json.load(open(os.path.join(GOLDEN, "golden.json")))[name]
    ref = np.load(os.path.join(GOLDEN, name + ".npy"))
    prog = oracle.OracleProgram(path=os.path.join(SCENES, name + ".gml"))
    img, _, cnt = prog.render(0, w=meta["w"], h=meta["h"], depth=meta["depth"], threads=4)
    assert (img == ref).all()
    assert cnt == meta["counts"]


def test_independent_python_restatement_agrees_bit_for_bit(oracle):
    # tests/independent_render.py restates the reference's path in plain Python with the scenes built by hand: no
    # shared code with the oracle or with the GML front-end that builds the oracle's scenes.  The two must give the
    # same unquantised f64 pixels (same operations in the same order), bytes and ray counts.
    import independent_render as ind
    w, h = 200, 150
    img, raw, cnt = ind.render(ind.build_spheres, w, h, 3, 90.0, (0.4, 0.4, 0.4))
    ref, ref64, ref_cnt = oracle.OracleProgram(path=os.path.join(SCENES, "spheres.gml")).render(0, w=w, h=h, want_f64=True, threads=4)
    assert (np.array(img, dtype=np.uint8) == ref).all()
    assert (np.array(raw) == ref64).all()
    assert cnt == {k: ref_cnt[k] for k in ("primary", "shadow", "reflect")}
    assert cnt["shadow"] > 0 and cnt["reflect"] > 0


def test_independent_python_restatement_all_primitives_rules_and_lights(oracle):
    # the same for a scene with a cube, sphere, cylinder, cone and two planes under scale / uscale / rotatex/y/z /
    # translate, joined by difference, intersect and union, lit by a directional, a point and a spot light, with a
    # surface function of face, u and v: the GML text goes to the oracle, the hand-built objects to the Python code
    import independent_render as ind
    w, h = 160, 120
    img, raw, cnt = ind.render(ind.build_mixed, w, h, 2, 80.0, (0.2, 0.25, 0.3))
    ref, ref64, ref_cnt = oracle.OracleProgram(text=ind.MIXED_GML % (w, h)).render(0, want_f64=True, threads=4)
    got, raw = np.array(img, dtype=np.uint8), np.array(raw)
    print("differing bytes", int((got != ref).sum()), "max |f64 diff|", float(np.abs(raw - ref64).max()), cnt)
    assert cnt == {k: ref_cnt[k] for k in ("primary", "shadow", "reflect")}
    assert (raw == ref64).all()
    assert (got == ref).all()
    assert cnt["shadow"] > 0 and cnt["reflect"] > 0 and len(np.unique(got)) > 50


def test_independent_python_restatement_chess_towers_with_surface_written_in_python(oracle):
    # two towers of gml/chess.gml on its board, nested union / intersect / difference with exactly shared surfaces
    # (gml/chess.gml:76-104), the procedural figureSurface and the board's checker written directly in Python
    # (gml/chess.gml:28-60) instead of going through any GML evaluator: front-end errors that the oracle and the
    # product would share (a transform builder, `modi`, `asin`, the order of operands) show up here
    import independent_render as ind
    w, h, depth = 120, 90, 3
    img, raw, cnt = ind.render(ind.build_towers, w, h, depth, 60.0, (0.33, 0.33, 0.33))
    ref, ref64, ref_cnt = oracle.OracleProgram(text=ind.TOWER_GML % (depth, w, h)).render(0, want_f64=True, threads=4)
    got, raw = np.array(img, dtype=np.uint8), np.array(raw)
    print("differing bytes", int((got != ref).sum()), "max |f64 diff|", float(np.abs(raw - ref64).max()), cnt)
    assert cnt == {k: ref_cnt[k] for k in ("primary", "shadow", "reflect")}
    assert (raw == ref64).all() and (got == ref).all()
    assert cnt["reflect"] > 2000 and len(np.unique(got.reshape(-1, 3), axis=0)) > 100


def test_tie_order_assumed_stable_obj1_first_also_beyond_twenty_events(oracle):
    # 24 events in one operator, every distance twice (two coinciding rows of spheres): the first of two tied
    # events decides whose colour a pixel gets.  The order-preserving sort is an ASSUMPTION about the reference's
    # `sort_unstable_by` (primitives.rs:271) that holds for Rust's insertion sort up to 20 elements; beyond that the
    # reference's order is unspecified.  What is checked here is that the three restatements share that one rule.
    import independent_render as ind
    w, h = 120, 60
    ind.MAX_EVENTS_SEEN[0] = 0
    img, raw, cnt = ind.render(ind.build_ties, w, h, 2, 60.0, (0.3, 0.3, 0.3))
    assert ind.MAX_EVENTS_SEEN[0] > 20
    ref, ref64, ref_cnt = oracle.OracleProgram(text=ind.TIES_GML % (w, h)).render(0, want_f64=True, threads=4)
    got = np.array(img, dtype=np.uint8)
    assert cnt == {k: ref_cnt[k] for k in ("primary", "shadow", "reflect")}
    assert (np.array(raw) == ref64).all() and (got == ref).all()
    # obj1 (the red row) wins every tie: no blue pixel anywhere, although the blue row is all there
    rgb = got.astype(np.int32)
    reddish = (rgb[..., 0] > rgb[..., 2] + 40).sum()
    bluish = (rgb[..., 2] > rgb[..., 0] + 40).sum()
    assert reddish > 50 and bluish == 0
