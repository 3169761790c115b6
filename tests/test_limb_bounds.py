"""The lazily reduced 29-bit-limb arithmetic is only safe while no 64-bit column sum can overflow and every point an
addition returns satisfies the invariant the next one assumes; an overflow would be silent for random inputs.
tools/fq29_bounds.py re-derives the bounds of the formulas in csrc/bn254_ec29.hip.h by interval arithmetic."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("fq29_bounds", os.path.join(ROOT, "tools", "fq29_bounds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_formula_stays_inside_its_limb_bounds():
    fb = _load()
    assert fb.main() == 0, fb.problems


def test_the_verifier_catches_a_violation():
    """Sanity of the checker itself: two un-normalised (2^31) operands of one product must be reported."""
    fb = _load()
    fb.problems.clear()
    wide = fb.Fe([(1 << 31) - 1] * 8 + [5], 17.0, "wide")
    fb.mul(wide, wide, "bad")
    assert any("can reach" in p for p in fb.problems)
    fb.problems.clear()
    fb.sub("K4E30", fb.zero(), wide, "neg")        # subtrahend limbs above the lift
    assert any("exceed the lift" in p for p in fb.problems)
