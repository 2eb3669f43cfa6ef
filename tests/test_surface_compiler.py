"""Surface closures compiled to straight-line bytecode vs the tree-walking evaluator
(the reference's per-hit protocol, evaluator.rs:776-792).  Bit-exact agreement is required."""
import ctypes
import os

import pytest

from conftest import SCENES


@pytest.mark.parametrize("name", ["spheres", "spotlight", "house", "cone-fractal", "chess", "fractal", "large"])
def test_scene_surfaces_bit_exact(helpers, name):
    stats = (ctypes.c_double * 5)()
    err = ctypes.create_string_buffer(512)
    n = helpers.lib.th_surface_compare(os.path.join(SCENES, name + ".gml").encode(), 0, 200, 7, stats, err, 512)
    assert n > 0, err.value
    comparisons, exact, maxdiff = stats[0], stats[1], stats[2]
    assert comparisons == n * 200 * 6 and exact == comparisons and maxdiff == 0.0


def _eval(helpers, text, var, face, u, v):
    ref = (ctypes.c_double * 6)()
    got = (ctypes.c_double * 6)()
    is_const, n_code = ctypes.c_int(), ctypes.c_int()
    err = ctypes.create_string_buffer(512)
    rc = helpers.lib.th_surface_eval_text(text.encode(), var.encode(), face, u, v, ref, got, ctypes.byref(is_const),
                                          ctypes.byref(n_code), err, 512)
    return rc, list(ref), list(got), bool(is_const.value), n_code.value, err.value.decode()


def test_constant_surface_detected(helpers):
    rc, ref, got, is_const, n_code, _ = _eval(helpers, "0.3 /k { /v /u /face k k k point 1.0 0.0 1.0 } /s", "s", 0, 0.1, 0.2)
    assert rc == 0 and is_const and n_code == 0 and ref == got == [0.3, 0.3, 0.3, 1.0, 0.0, 1.0]


def test_reference_small_program_as_surface(helpers):
    # the chess-style program of evaluator.rs:952-997 as a surface: stripes by angle around (0.5,0.5)
    text = ("{ /y /x x x mulf y y mulf addf sqrt } /dist 0.1 0.2 0.3 point /col1 0.9 0.8 0.7 point /col2 "
            "{ /v /u /face u 0.5 subf /u v 0.5 subf /v u u v dist apply divf /b "
            "0.0 v lessf { b asin } { 360.0 b asin subf } if 180.0 addf 30.0 divf floor 2 modi 1 eqi "
            "{ col1 } { col2 } if 0.7 0.3 1.0 } /s")
    for ui in range(10):
        for vi in range(10):
            rc, ref, got, is_const, n_code, err = _eval(helpers, text, "s", 1, ui / 10.0, vi / 10.0)
            assert rc == 0, err
            assert not is_const and n_code > 5
            assert [repr(x) for x in ref] == [repr(x) for x in got], (ui, vi)


def test_data_dependent_array_get(helpers):
    # gml/checked-cube.gml / gml/cube.gml style: colour picked from an array by face
    text = ("[ 1.0 0.0 0.0 point 0.0 1.0 0.0 point 0.0 0.0 1.0 point 1.0 1.0 0.0 point 0.0 1.0 1.0 point 1.0 0.0 1.0 point ] /cols "
            "{ /v /u /face cols face get 1.0 0.0 1.0 } /s")
    for face in range(6):
        rc, ref, got, *_ = _eval(helpers, text, "s", face, 0.3, 0.6)
        assert rc == 0 and ref == got


def test_rebinding_and_snapshot_rule_in_surfaces(helpers):
    # a nested closure created after a rebinding sees the captured value when the name was captured (evaluator.rs:92-95)
    text = "0.25 /k { /v /u /face u /k { k } apply k v point 1.0 0.0 1.0 } /s"
    rc, ref, got, *_ = _eval(helpers, text, "s", 0, 0.5, 0.75)
    assert rc == 0 and ref == got == [0.25, 0.5, 0.75, 1.0, 0.0, 1.0]


def test_integer_and_trig_ops(helpers):
    text = ("{ /v /u /face face 3 addi 2 muli 5 subi 4 divi 3 modi real u mulf v frac u 90.0 mulf sin v clampf point "
            "u u mulf v v mulf addf sqrt u cos v acos } /s")
    for (f, u, v) in [(0, 0.3, 0.9), (5, -1.7, 2.4), (2, 0.0, 0.0), (1, 3.14, -0.5)]:
        rc, ref, got, *_ = _eval(helpers, text, "s", f, u, v)
        assert rc == 0 and [repr(x) for x in ref] == [repr(x) for x in got]


def test_unsupported_surfaces_are_reported(helpers):
    # arms that leave different closures cannot be joined
    rc, *_, err = _eval(helpers, "{ /v /u /face 0.0 u lessf { { 1.0 } } { { 2.0 } } if apply u v point 1.0 0.0 1.0 } /s",
                        "s", 0, 0.5, 0.5)
    assert rc == -2 and err.startswith("UNSUPPORTED")
    # unbounded recursion on a data-dependent condition
    rc, *_, err = _eval(helpers, "{ /self /x x 0.0 lessf { x } { x 1.0 subf self self apply } if } /f "
                                 "{ /v /u /face u f f apply v v point 1.0 0.0 1.0 } /s", "s", 0, 5.0, 0.5)
    assert rc == -2 and err.startswith("UNSUPPORTED")


def _traps(helpers, text, var, face, u, v):
    got = (ctypes.c_double * 6)()
    ref_failed = ctypes.c_int()
    err = ctypes.create_string_buffer(512)
    helpers.lib.th_surface_traps.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_longlong, ctypes.c_double,
                                             ctypes.c_double, ctypes.POINTER(ctypes.c_int),
                                             ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
    rc = helpers.lib.th_surface_traps(text.encode(), var.encode(), face, u, v, ctypes.byref(ref_failed), got, err, 512)
    return rc, bool(ref_failed.value), list(got), err.value.decode()


def test_run_time_failures_raise_the_trap_exactly_where_the_reference_aborts(helpers):
    # evaluator.rs:357 (divi by zero), :422 (modi), :544-555 (get out of range) abort the reference at run time.
    # The straight-line bytecode computes both arms of an `if`, so the trap carries the path condition: it is
    # raised where the tree-walking evaluator throws and nowhere else.
    cases = {
        # k = floor(u) is zero for u in [0, 1)
        "div": "{ /v /u /face 6 u floor divi real 0.1 mulf /c c c c point 1.0 0.0 1.0 } /s",
        "mod": "{ /v /u /face 7 u floor modi real 0.1 mulf /c c c c point 1.0 0.0 1.0 } /s",
        # the same division guarded by an `if`: never fails, whatever u
        "guarded": "{ /v /u /face u floor /k k 0 eqi { 0 } { 6 k divi } if real 0.1 mulf /c c c c point 1.0 0.0 1.0 } /s",
        # the division sits on the arm taken only for k = 0: fails exactly there
        "wrong-guard": "{ /v /u /face u floor /k k 0 eqi { 6 k divi } { 1 } if real 0.1 mulf /c c c c point 1.0 0.0 1.0 } /s",
        # array of three indexed by floor(u): fails outside 0..2
        "get": "[ 0.1 0.5 0.9 ] /a { /v /u /face a u floor get /c c c c point 1.0 0.0 1.0 } /s",
        # nested: fails only for face 1 and k = 0
        "nested": "{ /v /u /face u floor /k face 1 eqi { k 0 lessi { 1 } { 5 k modi } if } { 2 } if real /c c c c point 1.0 0.0 1.0 } /s",
    }
    for name, text in cases.items():
        for face in (0, 1):
            for u in (-2.5, -0.5, 0.0, 0.5, 1.0, 2.5, 3.5):
                rc, ref_failed, got, err = _traps(helpers, text, "s", face, u, 0.25)
                assert rc in (0, 1), (name, err)
                assert bool(rc) == ref_failed, (name, face, u, err)
    # spot checks of the expectation itself
    assert _traps(helpers, cases["div"], "s", 0, 0.5, 0.0)[0] == 1 and _traps(helpers, cases["div"], "s", 0, 1.5, 0.0)[0] == 0
    assert all(_traps(helpers, cases["guarded"], "s", 0, u, 0.0)[0] == 0 for u in (0.0, 0.5, 3.0))
    assert _traps(helpers, cases["get"], "s", 0, 3.0, 0.0)[0] == 1 and _traps(helpers, cases["get"], "s", 0, 2.9, 0.0)[0] == 0
    assert _traps(helpers, cases["nested"], "s", 1, 0.5, 0.0)[0] == 1 and _traps(helpers, cases["nested"], "s", 0, 0.5, 0.0)[0] == 0


def test_a_failure_on_every_call_is_still_a_compile_time_error(helpers):
    rc, _, _, err = _traps(helpers, "{ /v /u /face 1 0 divi real /c c c c point 1.0 0.0 1.0 } /s", "s", 0, 0.5, 0.5)
    assert rc == -1 and "divide by zero" in err
