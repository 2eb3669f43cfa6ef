"""Host front-end: the reference's own unit tests restated against the C++ front-end.

tokenizer  /root/reference/src/tokenizer.rs:485-574
parser     /root/reference/src/parser.rs:78-144
evaluator  /root/reference/src/evaluator.rs:854-1058
plus the quirks the scenes depend on (no-op negi/negf, snapshot precedence, include guards).
"""
import math
import os

import pytest

from conftest import SCENES


# ---------------------------------------------------------------- tokenizer.rs:507-574
@pytest.mark.parametrize("text,expect", [
    ("1 % apa", "Integer(1)"),
    ("1 % apa\n2", "Integer(1) Integer(2)"),
    ("1", "Integer(1)"),
    ("123", "Integer(123)"),
    ("-1", "Integer(-1)"),
    ("-123", "Integer(-123)"),
    ("1 2", "Integer(1) Integer(2)"),
    ("123 321", "Integer(123) Integer(321)"),
    ("-1-1", "Integer(-1) Integer(-1)"),
    ("1.0", "Real(1)"),
    ("-1.0", "Real(-1)"),
    ("1.0e12", "Real(1000000000000)"),
    ("1e12", "Real(1000000000000)"),
    ("1e-12", "Real(9.9999999999999998e-13)"),
    ('"test"', "Str(test)"),
    ("true", "Boolean(true)"),
    ("false", "Boolean(false)"),
    ("/x", "Binder(x)"),
    ("/x-y_2", "Binder(x-y_2)"),
    ("x", "Identifier(x)"),
    ("x-y_2", "Identifier(x-y_2)"),
    ("addi", "Operator(addi)"),
    ("addiblaj", "Identifier(addiblaj)"),
    ("[1 2]", "[ Integer(1) Integer(2) ]"),
    ("{1 2}", "{ Integer(1) Integer(2) }"),
    ("{1 [2 3]}", "{ Integer(1) [ Integer(2) Integer(3) ] }"),
])
def test_tokenizer_reference_cases(helpers, text, expect):
    assert helpers.tokenize(text) == expect


def test_tokenizer_quirks(helpers):
    # a boolean fires on the bare prefix, before identifiers (tokenizer.rs:155-163, 455-456)
    assert helpers.tokenize("trueColor") == "Boolean(true) Identifier(Color)"
    # scanning stops silently at the first unrecognised character (tokenizer.rs:477-479)
    assert helpers.tokenize("1 2 ? 3") == "Integer(1) Integer(2)"
    # binding an operator name is not a token (tokenizer.rs:301-312): scanning stops there
    assert helpers.tokenize("1 /addi 2") == "Integer(1)"
    # block comments (tokenizer.rs:108-121)
    assert helpers.tokenize("/* a\n b */ 7") == "Integer(7)"
    # a real needs digits on both sides of the point
    assert helpers.tokenize("2.5 -0.125 addf") == "Real(2.5) Real(-0.125) Operator(addf)"
    # all fifty operator keywords (tokenizer.rs:199-253)
    ops = ("acos addf addi apply asin clampf cone cos cube cylinder difference divf divi eqf eqi floor frac get "
           "getx gety getz if intersect length lessf lessi light modi mulf muli negf negi plane point pointlight "
           "real render rotatex rotatey rotatez scale sin sphere spotlight sqrt subf subi translate union uscale").split()
    assert len(ops) == 50
    assert helpers.tokenize(" ".join(ops)) == " ".join("Operator(%s)" % o for o in ops)


# ---------------------------------------------------------------- parser.rs:78-144
def test_parser_nesting(helpers):
    assert helpers.parse("[1 2]") == "Array(Integer(1) Integer(2))"
    assert helpers.parse("{1 2}") == "Function(Integer(1) Integer(2))"
    assert helpers.parse("{1 [2 3]}") == "Function(Integer(1) Array(Integer(2) Integer(3)))"


@pytest.mark.parametrize("text", ["{", "[", "{1 [2 3 } ]", "} 1", "] 1"])
def test_parser_syntax_errors(helpers, text):
    assert helpers.parse(text) == "ERROR SyntaxError"


def test_parser_trailing_close_quirk(helpers):
    # do_parse returns at any closing token; a stray one that is the LAST token slips through (parser.rs:33-38, 70-75)
    assert helpers.parse("1 }") == "Integer(1)"


# ---------------------------------------------------------------- evaluator.rs:854-1058
def test_bind(helpers):
    assert helpers.run("1 /x", "x") == " | Integer(1)"
    assert helpers.run('"apa" /x', "x") == " | Str(apa)"


@pytest.mark.parametrize("text,expect", [
    ("1 2 addi", "Integer(3)"),
    ("1.5 2.5 addf", "Real(4)"),
    ("1 /x x x addi", "Integer(2)"),
    ("1 { /x x x } apply", "Integer(1) Integer(1)"),
    ("true { 1 } { 2 } if", "Integer(1)"),
    ("false { 1 } { 2 } if", "Integer(2)"),
    ("1 2 eqi", "false"), ("5 5 eqi", "true"),
    ("1.5 2.7 eqf", "false"), ("5.123 5.123 eqf", "true"),
    ("2 1 lessi", "false"), ("2 2 lessi", "false"), ("1 2 lessi", "true"),
    ("2.0 1.0 lessf", "false"), ("2.0 2.0 lessf", "false"), ("1.0 2.0 lessf", "true"),
])
def test_scalar_ops(helpers, text, expect):
    assert helpers.run(text) == expect


def test_successful_run(helpers):
    for prog in ["false /b b { 1 } { 2 } if", "4 /x 2 x addi", "1 { /x x x } apply addi",
                 "{ /x x x } /dup { dup apply muli } /sq 3 sq apply", "{ /x /y x y } /swap 3 4 swap apply"]:
        assert not helpers.run(prog).startswith("ERROR")
    fact = "{ /self /n n 2 lessi { 1 } { n 1 subi self self apply n muli } if } /fact 12 fact fact apply"
    assert helpers.run(fact) == "Integer(479001600)"


def _small_program_expected(u, v):
    # the Rust restatement inside test_small_program (evaluator.rs:970-989)
    u -= 0.5
    v -= 0.5
    b = u / math.sqrt(u * u + v * v)
    c = math.degrees(math.asin(b)) if 0.0 < v else 360.0 - math.degrees(math.asin(b))
    ci = math.floor((c + 180.0) / 30.0)
    ci = ci - 2 * int(ci / 2)  # modi: remainder with the sign of the dividend
    return 1 if ci == 1 else 2


def test_small_program_grid(helpers):
    for ui in range(10):
        for vi in range(10):
            if ui == 5 and vi == 5:
                continue  # 0/0: asin(NaN); floor(NaN) as i64 is 0 in Rust, Python's floor raises
            u, v = ui / 10.0, vi / 10.0
            prog = ("1 /col1 2 /col2 %.1f /u %.1f /v { /y /x x x mulf y y mulf addf sqrt } /dist \n { \n"
                    " u 0.5 subf /u v 0.5 subf /v \n u u v dist apply divf /b \n 0.0 v lessf { b asin } "
                    "{ 360.0 b asin subf } if 180.0 addf 30.0 divf \n floor 2 modi 1 eqi { col1 } { col2 } if \n } apply"
                    % (u, v))
            assert helpers.run(prog) == "Integer(%d)" % _small_program_expected(u, v), (u, v)
    # the NaN corner: b = 0/0, asin(NaN) ... floor -> 0, 0 modi 2 = 0 -> col2
    prog_nan = ("1 /col1 2 /col2 0.5 /u 0.5 /v { /y /x x x mulf y y mulf addf sqrt } /dist { u 0.5 subf /u v 0.5 subf /v "
                "u u v dist apply divf /b 0.0 v lessf { b asin } { 360.0 b asin subf } if 180.0 addf 30.0 divf "
                "floor 2 modi 1 eqi { col1 } { col2 } if } apply")
    assert helpers.run(prog_nan) == "Integer(2)"


def test_clampf(helpers):
    assert helpers.run("-0.4 clampf /x", "x") == " | Real(0)"
    assert helpers.run("1.1 clampf /x", "x") == " | Real(1)"
    assert helpers.run("0.8 clampf /x", "x") == " | Real(0.80000000000000004)"


def test_array(helpers):
    assert helpers.run("[1 2 3] length /x", "x") == " | Integer(3)"
    assert helpers.run("[1 2 3] 1 get /x", "x") == " | Integer(2)"
    assert helpers.run("[0 1 2 3 4 5 6 7 8 9] /a a 0 get /x", "x") == " | Integer(0)"
    assert helpers.run("[1 2 3] 3 get").startswith("ERROR")   # off-by-one bound check then index panic (evaluator.rs:549-553)
    assert helpers.run("[1 2 3] -1 get").startswith("ERROR")


def test_point(helpers):
    assert helpers.run("1.0 2.0 3.0 point /x", "x") == " | Point(1,2,3)"
    assert helpers.run("1.0 2.0 3.0 point getx /x", "x") == " | Real(1)"
    assert helpers.run("1.0 2.0 3.0 point gety /x", "x") == " | Real(2)"
    assert helpers.run("1.0 2.0 3.0 point getz /x", "x") == " | Real(3)"


def test_sphere_and_light(helpers):
    assert helpers.run("1.0 { /x x } sphere /x", "x") == "Real(1) | Node"
    # evaluator.rs:1044-1053: direction = normalize(-d), intensity = colour
    got = helpers.run("1.0 0.0 0.0 point 0.7 0.5 0.3 point light /x", "x")
    assert got == " | Light(kind=0,dir=-1,-0,-0,color=0.69999999999999996,0.5,0.29999999999999999)"


def test_render_hook(helpers):
    prog = ('1.0 0.0 0.0 point 0.7 0.5 0.3 point light /l 1.0 { /v /u /face 0.8 0.2 v point 1.0 0.2 1.0 } sphere /s '
            '0.5 0.5 0.5 point [ l ] s 3 90.0 320 240 "eval.ppm" render')
    assert helpers.run(prog) == "Real(1) | renders=1"


# ---------------------------------------------------------------- quirks kept on purpose
def test_neg_ops_are_noops(helpers):
    # evaluator.rs:446-456
    assert helpers.run("5 negi") == "Integer(5)"
    assert helpers.run("2.5 negf") == "Real(2.5)"


def test_degrees(helpers):
    assert helpers.run("90.0 sin") == "Real(1)"
    assert helpers.run("90.0 cos") == "Real(0)"        # |r| < 1e-15 snapped (evaluator.rs:347-355)
    assert helpers.run("180.0 sin") == "Real(0)"
    assert helpers.run("1.0 asin") == "Real(90)"
    assert helpers.run("0.0 acos") == "Real(90)"


def test_integer_ops(helpers):
    assert helpers.run("-7 2 divi") == "Integer(-3)"   # truncating
    assert helpers.run("-7 2 modi") == "Integer(-1)"   # sign of the dividend (evaluator.rs:216-218)
    assert helpers.run("7 0 divi").startswith("ERROR")
    assert helpers.run("-1.5 floor") == "Integer(-2)"
    assert helpers.run("-1.25 frac") == "Real(-0.25)"  # fract keeps the sign
    assert helpers.run("3 real") == "Real(3)"


def test_snapshot_precedence(helpers):
    # SnapMap::snapshot lets the OLD snapshot overwrite newer bindings (evaluator.rs:92-95): inside a closure that
    # captured x=1, rebinding x and then creating a closure gives that closure the captured x, not the rebound one.
    assert helpers.run("1 /x { 2 /x { x } apply } apply") == "Integer(1)"
    # while a direct read sees the rebinding (current map first, evaluator.rs:108-113)
    assert helpers.run("1 /x { 2 /x x } apply") == "Integer(2)"
    # at top level the snapshot is empty, so later closures see later bindings
    assert helpers.run("1 /x 2 /x { x } apply") == "Integer(2)"


def test_if_arm_bindings_do_not_leak(helpers):
    assert helpers.run("1 /x true { 2 /x } { } if x") == "Integer(1)"


def test_array_shares_environment(helpers):
    # arrays evaluate in the enclosing environment (evaluator.rs:804-808)
    assert helpers.run("[ 1 /x ] length x") == "Integer(0) Integer(1)"


def test_lights_array_is_consumed_reversed(helpers):
    # make_array stores reversed (evaluator.rs:249-253); render walks the storage order (evaluator.rs:531-542).
    assert helpers.run("[1 2 3]") == "[Integer(1) Integer(2) Integer(3)]"


def test_preprocessor_guards(helpers):
    # rungml/src/main.rs:17-47: "#endif " decrements unconditionally, so after a first-time include the tail of that
    # file is dropped; a second include of a guarded file contributes nothing; the includer goes on normally.
    text = helpers.preprocess(os.path.join(SCENES, "spheres.gml"))
    assert text.count("/matte") == 1 and text.count("/dup") == 1 and text.count("/white") == 1
    assert "#include" not in text and "#ifndef" not in text and "#endif" not in text
    assert "/* surface.ins" in text                      # block comments reach the tokenizer
    assert text.rstrip().endswith("render")
    text2 = helpers.preprocess(os.path.join(SCENES, "cone-fractal.gml"))  # includes colors.ins twice
    assert text2.count("/magenta") == 1


def test_all_config_scenes_evaluate(helpers):
    for name in ["spheres", "chess", "fractal", "cone-fractal", "spotlight", "house", "large"]:
        text = helpers.preprocess(os.path.join(SCENES, name + ".gml"))
        out = helpers.run(text)
        assert out.endswith("renders=1"), (name, out[:200])
