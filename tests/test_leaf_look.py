"""The look before a leaf test (prim_may_enter, gmlrt_prims.hpp) against the five intersectors, on the host: whenever the
look says that a ray cannot have an Entry on the primitive, the intersector (the restatement of primitives.rs:349-981
that the kernel runs) must report none.  The kernel skips the intersector for a leaf when the look turns every lane of
the wave away, so a wrong "no" here would be a wrong pixel.  tests/native/look_fuzz.cpp draws random, lattice, grazing
and near-surface rays; the hardware's approximate reciprocal and square root are stood in for by exact ones with up to
2^-22 of relative error of either sign."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def look_fuzz(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("look") / "look_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "look_fuzz.cpp")])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_look_never_turns_away_a_ray_with_an_entry(look_fuzz, seed):
    p = subprocess.run([look_fuzz, "150000", str(seed)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if "violations" in l]
    assert len(lines) == 5 and all(l.endswith(" 0 violations") for l in lines), p.stdout[-2000:]
    # and it is worth its cost: it turns away most rays that have no Entry
    for l in lines:
        share = float(l.split("(")[1].split("%")[0])
        assert share > 50.0, l
