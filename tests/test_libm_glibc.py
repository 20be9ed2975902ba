"""picasso_amd/csrc/libm_glibc.h against the C library it restates (CPU tier: the header compiles for the host).

The strict MLE kernel evaluates math.erf / math.exp of picasso/gaussmle.py:279, 295, 313, 357 with these functions; here they
are compared bit for bit with glibc's on the machine that runs the test (the oracle calls the same library).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "picasso_amd", "csrc")


def _glibc_version():
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.gnu_get_libc_version.restype = ctypes.c_char_p
        return tuple(int(v) for v in libc.gnu_get_libc_version().decode().split(".")[:2])
    except Exception:
        return None


pytestmark = pytest.mark.skipif(_glibc_version() is None or _glibc_version() < (2, 28) or os.uname().machine != "x86_64",
                                reason="the header restates glibc >= 2.28 on x86-64")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("libm") / "libm_glibc_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, "-o", so,
                    os.path.join(ROOT, "tests", "native", "libm_glibc_host.cpp")], check=True)
    lib = ctypes.CDLL(so)
    for f in (lib.cmp_exp, lib.cmp_erf):
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    return lib


def _differing(fn, x):
    x = np.ascontiguousarray(x, np.float64)
    first = ctypes.c_int64(-1)
    bad = fn(x.ctypes.data, x.size, ctypes.byref(first))
    return bad, (float(x[first.value]).hex() if first.value >= 0 else None)


def _neighbours(values, width=3):
    out = []
    for v in values:
        w_up = w_dn = np.float64(v)
        out += [w_up, -w_up]
        for _ in range(width):
            w_up = np.nextafter(w_up, np.inf)
            w_dn = np.nextafter(w_dn, -np.inf)
            out += [w_up, -w_up, w_dn, -w_dn]
    return np.array(out)


def test_exp_table_is_its_formula():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_glibc_exp_table.py")], check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(CSRC, "libm_glibc_exp_table.inc")).read()


def test_exp_has_glibc_bits(host):
    rng = np.random.default_rng(2028)
    n = 2_000_000
    cases = {
        "whole range": rng.uniform(-750, 720, n),
        "the fit's arguments": -np.exp(rng.uniform(-45, 7, n)),
        "subnormal results": rng.uniform(-1100, -700, n),
        "overflow": rng.uniform(700, 1100, n // 4),
        "random bit patterns": rng.integers(0, 2 ** 64, n, dtype=np.uint64).view(np.float64),
        "special values": np.concatenate([
            [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-300, -1e-300],
            _neighbours([2.0 ** -54, 512.0, 1024.0, 709.782712893384, 745.1332191019411, 708.3964185322641])]),
    }
    for name, x in cases.items():
        assert _differing(host.cmp_exp, x) == (0, None), name


def test_erf_has_glibc_bits(host):
    rng = np.random.default_rng(2029)
    n = 2_000_000
    cases = {
        "all branches": rng.uniform(-7, 7, n),
        "the two rational ranges below 1.25": rng.uniform(-1.3, 1.3, n),
        "every magnitude": np.exp(rng.uniform(-720, 3, n)) * rng.choice([-1.0, 1.0], n),
        "random bit patterns": rng.integers(0, 2 ** 64, n, dtype=np.uint64).view(np.float64),
        "range ends and special values": np.concatenate([
            [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-310, 27.0, -27.0],
            _neighbours([0.84375, 1.25, 1 / 0.35, 6.0, 2.0 ** -28, 2.0 ** -1015])]),
    }
    for name, x in cases.items():
        assert _differing(host.cmp_erf, x) == (0, None), name
