"""abd_terms.hpp on the CPU: the per-variable form of the closed-form terms (what a leapfrog train's launch evaluates on the
device, one lane per value variable) equals the serial form that assembles every fetched evaluation -- which the GPU
parity tests pin against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("terms") / "libterms_harness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility-inlines-hidden", "-Wl,-Bsymbolic", "-I", os.path.join(ROOT, "abdpymc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "terms_harness.cpp"), "-o", str(out)])
    lib = C.CDLL(str(out))
    dp = C.POINTER(C.c_double)
    lib.terms_compare.argtypes = [dp, dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp]
    lib.terms_compare.restype = C.c_int
    return lib


@pytest.mark.parametrize("dense", [0, 1])
def test_per_variable_form_equals_the_serial_form(lib, dense):
    rng = np.random.default_rng(3 + dense)
    dp = C.POINTER(C.c_double)
    for trial in range(200):
        theta = rng.normal(size=17) * 1.5
        if trial % 7 == 0:
            theta[11] = 0.0  # b = 0: the guarded scale
        G, N = int(rng.integers(2, 300)), float(rng.integers(1, 20000))
        sums = rng.normal(size=16) * 10.0 ** rng.integers(-2, 6)
        sums[13], sums[14] = float(rng.integers(0, 1000)), float(rng.integers(0, int(N) + 1))
        lp_s, lp_l = C.c_double(), C.c_double()
        g_s, g_l = np.empty(17), np.empty(17)
        rc = lib.terms_compare(theta.ctypes.data_as(dp), sums.ctypes.data_as(dp), G, N, G * N, G * N, dense, C.byref(lp_s),
                               g_s.ctypes.data_as(dp), C.byref(lp_l), g_l.ctypes.data_as(dp))
        assert rc == 0
        assert abs(lp_s.value - lp_l.value) <= 1e-12 * max(1.0, abs(lp_s.value))
        np.testing.assert_allclose(g_l, g_s, rtol=1e-12, atol=1e-12 * np.abs(g_s).max())
