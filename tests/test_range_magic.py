"""The dense kernel cuts the (lane group, gap) plane into ranges with row / G by a 32-bit reciprocal (abd_dense.hpp: range_of).
abd_create keeps a cohort on the dense path only where abd_div_magic_exact says every quotient is exact: swept here on the
CPU at the largest shapes the predicate accepts, and shown to be needed just beyond them."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("magic") / "libmagic_harness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility-inlines-hidden", "-Wl,-Bsymbolic", "-I", os.path.join(ROOT, "abdpymc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "magic_harness.cpp"), "-o", str(out)])
    lib = C.CDLL(str(out))
    lib.magic_sweep.argtypes = [C.c_ulonglong, C.c_uint]
    lib.magic_sweep.restype = C.c_longlong
    lib.magic_exact.argtypes = [C.c_ulonglong, C.c_uint]
    lib.magic_exact.restype = C.c_int
    return lib


def largest_accepted(lib, G):
    lo, hi = 0, 1 << 31  # accepted at lo, refused at hi
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if lib.magic_exact(mid, G):
            lo = mid
        else:
            hi = mid
    return lo


@pytest.mark.parametrize("G", [2, 3, 7, 31, 60, 200, 257, 500, 511, 512])
def test_quotients_are_exact_wherever_the_dense_path_is_taken(lib, G):
    n_rows = largest_accepted(lib, G)
    assert n_rows >= 1 << 20
    assert lib.magic_sweep(n_rows, G) == 0


def test_the_predicate_is_not_idle(lib):
    # the advisor's case: 500 gaps, 4 000 000 individuals (62 500 lane groups) -- G * N < 2^31 but the reciprocal is wrong there
    G, n_rows = 500, 62500 * 500
    assert not lib.magic_exact(n_rows, G)
    assert lib.magic_sweep(n_rows, G) > 0
    # config 3 and config 5 are far inside
    assert lib.magic_exact(157 * 200, 200) and lib.magic_exact(1563 * 200, 200)
