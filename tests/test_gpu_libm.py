"""GPU tier: the device build of csrc/libm_glibc.h returns the bits of the host's C library (the one the oracle — and, under
numba, the reference's math.erf / math.exp, picasso/gaussmle.py:279, 295 — calls), through the C ABI (pmi_libm_eval_dev)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("libm") / "libm_glibc_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "picasso_amd", "csrc"),
                    "-o", so, os.path.join(ROOT, "tests", "native", "libm_glibc_host.cpp")], check=True)
    lib = ctypes.CDLL(so)
    for f in (lib.ref_exp, lib.ref_erf):
        f.restype = None
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    return lib


def _device(fn, x):
    import torch
    from picasso_amd import _lib
    xt = torch.from_numpy(np.ascontiguousarray(x, np.float64)).cuda()
    out = torch.empty_like(xt)
    with _lib.lock():
        _lib.check(_lib.load().pmi_libm_eval_dev(fn, ctypes.c_void_p(xt.data_ptr()), xt.numel(), ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pmi_libm_eval_dev")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _host(f, x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    f(x.ctypes.data, x.size, out.ctypes.data)
    return out


def _args(rng, n):
    edges = []
    for v in (0.84375, 1.25, 1 / 0.35, 6.0, 2.0 ** -28, 2.0 ** -54, 512.0, 1024.0, 709.782712893384, 745.1332191019411, 708.3964185322641):
        w_up = w_dn = np.float64(v)
        for _ in range(4):
            edges += [w_up, -w_up, w_dn, -w_dn]
            w_up, w_dn = np.nextafter(w_up, np.inf), np.nextafter(w_dn, -np.inf)
    return np.concatenate([
        rng.uniform(-7, 7, n), rng.uniform(-1.3, 1.3, n), rng.uniform(-750, 720, n), -np.exp(rng.uniform(-45, 7, n)),
        rng.uniform(-1100, -700, n // 4), np.exp(rng.uniform(-720, 3, n)) * rng.choice([-1.0, 1.0], n),
        rng.integers(0, 2 ** 64, n, dtype=np.uint64).view(np.float64),
        [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-310, -1e-310, 27.0, -27.0], edges])


@pytest.mark.parametrize("fn,name", [(0, "exp"), (1, "erf")])
def test_device_libm_has_the_host_librarys_bits(host, fn, name):
    x = _args(np.random.default_rng(40 + fn), 2_000_000)
    dev = _device(fn, x)
    ref = _host(host.ref_exp if fn == 0 else host.ref_erf, x)
    same = (dev.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(dev) & np.isnan(ref))
    assert same.all(), (name, int((~same).sum()), [float(v).hex() for v in x[~same][:5]])


def test_device_librarys_own_functions_differ_in_the_last_bit():
    """Why the header exists: the device library's exp / erf are faithful too, and not the same function."""
    from math import erf
    x = np.random.default_rng(7).uniform(-4, 4, 200_000)
    dev = _device(3, x)
    ref = np.array([erf(v) for v in x])
    differ = dev.view(np.uint64) != ref.view(np.uint64)
    assert 0 < differ.mean() < 0.5, differ.mean()
    assert np.max(np.abs(dev - ref) / np.maximum(np.abs(ref), 1e-300)) < 4.5e-16
