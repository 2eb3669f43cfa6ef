"""The acceleration hints in the scene blob must be conservative: every point of a subtree's solid lies
inside the sphere AND inside the box of every bound placed in front of that subtree.  Checked here by
sampling points on each leaf's surface, deciding by CSG point membership whether a point belongs to the
bounded subtree's solid, and testing the ones that do against both volumes.  (What a missed bound would
cost is a wrong image; the GPU parity tests would catch that only for the rays they happen to shoot.)"""
import os
import struct

import numpy as np
import pytest

from conftest import SCENES

LEAF_BYTES = 208
BOUND_BYTES = 80
TOL = 1e-9


def parse(blob):
    (magic, version, total, n_leaves, n_nodes, n_lights, n_code, n_consts, max_stack, max_events,
     n_bounds) = struct.unpack_from("<IIQIIIIIIII", blob, 0)
    off_leaves, off_nodes, off_lights, off_code, off_consts, off_bounds = struct.unpack_from("<6Q", blob, 48)
    assert version == 3
    leaves = []
    for i in range(n_leaves):
        f = struct.unpack_from("<24d", blob, off_leaves + i * LEAF_BYTES)
        typ = struct.unpack_from("<I", blob, off_leaves + i * LEAF_BYTES + 192)[0]
        inv_m = np.array(f[:12]).reshape(3, 4)
        m = np.array(f[12:]).reshape(3, 4)
        leaves.append((typ, inv_m, m))
    nodes = struct.unpack_from(f"<{n_nodes}I", blob, off_nodes)
    bounds = [struct.unpack_from("<10d", blob, off_bounds + i * BOUND_BYTES) for i in range(n_bounds)]
    return leaves, nodes, bounds


def surface_samples(typ, rng, n):
    """Points on the boundary of the unit primitive (primitives.rs:328, 441, 576, 754, 924)."""
    u, v = rng.random(n), rng.random(n)
    ang = 2 * np.pi * u
    if typ == 0:  # sphere
        z = 2 * v - 1
        r = np.sqrt(1 - z * z)
        return np.stack([r * np.cos(ang), z, r * np.sin(ang)], 1)
    if typ == 1:  # cube [0,1]^3: a random face
        p = rng.random((n, 3))
        axis = rng.integers(0, 3, n)
        side = rng.integers(0, 2, n).astype(float)
        p[np.arange(n), axis] = side
        return p
    if typ == 2:  # cylinder: side, bottom, top
        which = rng.integers(0, 3, n)
        rad = np.where(which == 0, 1.0, np.sqrt(v))
        y = np.where(which == 0, v, np.where(which == 1, 0.0, 1.0))
        return np.stack([rad * np.cos(ang), y, rad * np.sin(ang)], 1)
    if typ == 3:  # cone: side (radius = y) and base disc at y = 1
        which = rng.integers(0, 2, n)
        y = np.where(which == 0, v, 1.0)
        rad = np.where(which == 0, v, np.sqrt(rng.random(n)))
        return np.stack([rad * np.cos(ang), y, rad * np.sin(ang)], 1)
    # plane y = 0: a patch around the origin
    return np.stack([20 * (u - 0.5), np.zeros(n), 20 * (v - 0.5)], 1)


def inside(typ, q):
    """Closed point membership of the unit primitive, object-space points q[n,3]."""
    x, y, z = q[:, 0], q[:, 1], q[:, 2]
    if typ == 0:
        return x * x + y * y + z * z <= 1 + TOL
    if typ == 1:
        return ((q >= -TOL) & (q <= 1 + TOL)).all(1)
    if typ == 2:
        return (x * x + z * z <= 1 + TOL) & (y >= -TOL) & (y <= 1 + TOL)
    if typ == 3:
        return (x * x + z * z <= y * y + TOL) & (y >= -TOL) & (y <= 1 + TOL)
    return y <= TOL


def to_object(inv_m, p):
    return p @ inv_m[:, :3].T + inv_m[:, 3]


def check(blob, seed=1234):
    """Returns (points checked, largest distance by which a sphere misses a point, box misses)."""
    leaves, nodes, bounds = parse(blob)
    rng = np.random.default_rng(seed)
    per_leaf = 24 if len(leaves) > 100 else 200
    samples = []  # world-space surface points of each leaf
    for typ, inv_m, m in leaves:
        o = surface_samples(typ, rng, per_leaf)
        samples.append(o @ m[:, :3].T + m[:, 3])

    checked, worst, box_misses = 0, -np.inf, 0
    for i, w in enumerate(nodes):
        if w >> 30 != 2:
            continue
        idx, nwords = (w >> 15) & 0x7FFF, w & 0x7FFF
        sub = nodes[i + 1:i + 1 + nwords]
        sub_leaves = [x & 0x3FFFFFFF for x in sub if x >> 30 == 0]
        pts = np.concatenate([samples[l] for l in sub_leaves])
        owner = np.concatenate([np.full(len(samples[l]), l) for l in sub_leaves])
        # CSG membership of every point in the subtree's solid (a point counts as inside its own leaf;
        # a point on B's surface inside A is on the surface of A - B, so B's own points survive the
        # difference as well)
        st = []
        for x in sub:
            k = x >> 30
            if k == 0:
                l = x & 0x3FFFFFFF
                typ, inv_m, _ = leaves[l]
                mine = owner == l
                st.append((inside(typ, to_object(inv_m, pts)) | mine, mine))
            elif k == 1:
                (b, bown), (a, aown) = st.pop(), st.pop()
                rule = x & 3
                if rule == 0:
                    st.append((a | b, aown | bown))
                elif rule == 1:
                    st.append((a & b, aown | bown))
                else:
                    st.append((a & (~b | bown), aown | bown))
        assert len(st) == 1
        member = st[0][0]
        c = np.array(bounds[idx][0:3])
        r = bounds[idx][3]
        lo, hi = np.array(bounds[idx][4:7]), np.array(bounds[idx][7:10])
        p = pts[member]
        if len(p):
            d = np.sqrt(((p - c) ** 2).sum(1))
            worst = max(worst, (d - r).max())
            box_misses += int((~((p >= lo) & (p <= hi)).all(1)).sum())
            checked += len(p)
    return checked, worst, box_misses


@pytest.mark.parametrize("name", ["spheres", "chess", "fractal", "cone-fractal", "spotlight", "house", "large"])
def test_bounds_enclose_their_subtrees(gmlrt, name):
    prog = gmlrt.Program(path=os.path.join(SCENES, name + ".gml"))
    blob = prog.calls[0].blob
    checked, worst, box_misses = check(blob)
    n_bounds = struct.unpack_from("<I", blob, 44)[0]
    if n_bounds:
        assert checked > 0
        assert worst <= 0.0, f"{name}: a sphere misses a point of its subtree by {worst}"
        assert box_misses == 0


def test_the_check_notices_a_bound_that_is_too_small(gmlrt):
    prog = gmlrt.Program(path=os.path.join(SCENES, "chess.gml"))
    blob = bytearray(prog.calls[0].blob)
    off_bounds = struct.unpack_from("<Q", blob, 48 + 5 * 8)[0]
    r = struct.unpack_from("<d", blob, off_bounds + 3 * BOUND_BYTES + 24)[0]
    struct.pack_into("<d", blob, off_bounds + 3 * BOUND_BYTES + 24, 0.8 * r)
    assert check(bytes(blob))[1] > 0.0
    blob = bytearray(prog.calls[0].blob)
    hi_y = struct.unpack_from("<d", blob, off_bounds + 3 * BOUND_BYTES + 64)[0]
    lo_y = struct.unpack_from("<d", blob, off_bounds + 3 * BOUND_BYTES + 40)[0]
    struct.pack_into("<d", blob, off_bounds + 3 * BOUND_BYTES + 64, lo_y + 0.7 * (hi_y - lo_y))
    assert check(bytes(blob))[2] > 0


def test_surface_of_difference_is_bounded(gmlrt):
    """The cut surface of A - B belongs to B's boundary: it must still be inside A's volumes."""
    text = """
    { /v /u /face 1.0 0.0 0.0 point 1.0 0.0 1.0 } /s
    s sphere 2.0 uscale  s cube -0.5 -0.5 -0.5 translate 3.0 uscale 1.0 0.0 0.0 translate  difference
    0.0 0.0 5.0 translate
    s sphere 0.5 uscale 4.0 4.0 9.0 translate union /scene
    0.5 0.5 0.5 point [ ] scene 1 90.0 8 8 "t.ppm" render
    """
    prog = gmlrt.Program(text=text)
    leaves, nodes, bounds = parse(prog.calls[0].blob)
    assert len(bounds) >= 1
    for b in bounds:
        assert np.isfinite(b[3]) and all(np.isfinite(b[4:10]))
