"""The step kernel divides by constants in three operations (sf_kernels.hip: sf_div_const / SF_DIV).  That is only
allowed because the result is the IEEE quotient bit for bit; this keeps a host run of that check in the suite
(4e6 operands per divisor here; 4e8 per divisor were run once when the change was made, none differed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constant_division_is_the_ieee_quotient(tmp_path):
    exe = str(tmp_path / "div_const")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "div_const.c"),
                           "-o", exe, "-lm"])
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("mismatches=0 of") == 9, out.stdout


def test_every_divisor_of_the_kernel_is_covered():
    """SF_DIV call sites in the kernel source use only divisors the host check covers."""
    import re
    src = open(os.path.join(ROOT, "spacefortress_amd", "csrc", "sf_kernels.hip")).read()
    lay = open(os.path.join(ROOT, "spacefortress_amd", "csrc", "sf_layout.h")).read()
    known = {"M_PI": None, "10": 10, "180": 180, "360": 360, "SF_MAX_MISSILES_D": 20, "sfc::ndist_b": 80, "sfc::pb_width": 90,
             "sfc::pb_height": 92, "sfc::max_ticks": 5294, "sfc::sector_size": 10}
    divisors = set()
    for m in re.finditer(r"SF_DIV\((.*)\)", src):
        if "define" in src[max(0, m.start() - 10):m.start()]:
            continue
        arg = m.group(1)
        depth, cut = 0, None
        for i, ch in enumerate(arg):  # the divisor is what follows the last top-level comma
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                cut = i
            if depth < 0:
                arg = arg[:i]
                break
        divisors.add(arg[cut + 1:].strip())
    assert divisors and divisors <= set(known), divisors - set(known)
    assert "ndist_b = (200 - 40) / 2.0" in lay and "pb_width = 90, pb_height = 92, max_ticks = 5294" in lay
    assert "sector_size = 10" in lay and "#define SF_MAX_MISSILES_D 20.0" in src


def test_near_axis_atan2_is_glibcs(tmp_path):
    """sf_atan2's near-axis form (sf_kernels.hip) against the host libm the reference engine runs on."""
    exe = str(tmp_path / "atan2_axis")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "atan2_axis.c"),
                           "-o", exe, "-lm"])
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_table_step_atan2_is_within_1e_15_of_libm(tmp_path):
    """sf_atan2_core (sf_kernels.hip), the hot path's atan2 -- reciprocal + Newton quotient, one table step, five series
    terms -- restated on the host with a float-precision reciprocal seed: within 1e-15 rad (2 ulps) of the host libm on
    position differences, velocities, the spawn lattice and the table's knots; axes and diagonals to the ulp."""
    exe = str(tmp_path / "atan2_core")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "atan2_core.c"),
                           "-o", exe, "-lm"])
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert out.returncode == 0 and "off by more than 1e-15 rad: 0 of" in out.stdout, out.stdout
    # the kernel's function is the one restated: same series coefficients, same table step
    src = open(os.path.join(ROOT, "spacefortress_amd", "csrc", "sf_kernels.hip")).read()
    for needle in ("rint(q * 16.0)", "__builtin_fma(s, 1.0 / 9.0, -1.0 / 7.0)", "__builtin_fma(q, c, 1.0)"):
        assert needle in src, needle
