"""The -DMSM_AMD_EXPERIMENTS build (build_ab/libmsm_amd_exp.so, made by __graft_entry__.build()): what was built, measured
and NOT shipped still has to be right -- the rejected multiplication forms (ops 32..36) and the accumulate-kernel
builds behind MSM_AMD_ACC_VARIANT (3 = three waves per SIMD, 2 = register-lean product scanning at four, 5 = the
compiler's column form at four with Y / ZZ / ZZZ parked in LDS, 4 = the hand-allocated five-wave statement of
tools/gen_accumulate_asm.py + its redo pass).  The library under test is chosen when the package is imported, so every
case runs the regular tests in a child process with MSM_AMD_LIB pointing at the experiments build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "build_ab", "libmsm_amd_exp.so")


def _child(args, extra_env, timeout):
    if not os.path.exists(EXP):
        pytest.skip("build_ab/libmsm_amd_exp.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = dict(os.environ, MSM_AMD_LIB=EXP, **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_generator_selftest_of_the_hand_allocated_kernel():
    """tools/gen_accumulate_asm.py --selftest: the emitted instruction stream of a point addition, run through the
    generator's own interpreter, against big integers (limbs and group elements)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_accumulate_asm.py"), "--selftest"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr


def test_unshipped_multiplication_forms_on_the_host():
    _child(["tests/test_host_fe29.py", "-k", "unshipped", "-m", "not gpu"], {}, 300)


@pytest.mark.gpu
def test_unshipped_multiplication_forms_on_the_device():
    _child(["tests/test_gpu_unit_ops.py", "-k", "unshipped", "-m", "gpu"], {}, 300)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [3, 2, 5, 4, 7, 8])
def test_accumulate_kernel_variants_agree_with_the_oracle(variant):
    """Whole MSMs (sizes 1 .. 1000, every window size, skewed and edge inputs incl. identity bases and cancelling
    points -- the redo pass of variant 4 --, the table pipeline) through each experimental accumulate kernel."""
    _child(["tests/test_gpu_msm.py", "-m", "gpu", "-k",
            "small or window_sizes or edge_scalars or skewed or zero_scalars or known_answers or precomputed_window or doubling_inside"],
           {"MSM_AMD_ACC_VARIANT": str(variant)}, 600)
