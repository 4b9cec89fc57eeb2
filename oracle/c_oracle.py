"""ctypes wrapper of oracle/libabd_oracle.so (plain-C CPU restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libabd_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.abd_oracle_logp_dlogp.restype = C.c_int
        _lib.abd_oracle_gibbs_sweep.restype = C.c_int
        _lib.abd_oracle_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return load().abd_oracle_max_threads()


class COracle:
    """Holds one cohort's arrays in the C layouts so repeated evaluations carry no conversion cost."""

    def __init__(self, cohort, splits=None, ignore_pcrpos=False):
        self.lib = load()
        self.G, self.N = int(cohort.n_gaps), int(cohort.n_inds)
        self.splits = np.asarray(tuple(splits or ()), dtype=np.int32)
        self.vacs = np.ascontiguousarray(cohort.vacs, dtype=np.int8)
        self.pcr = None if ignore_pcrpos else np.ascontiguousarray(cohort.pcrpos, dtype=np.int8)
        self.obs = []
        for o in (cohort.s, cohort.n):
            self.obs.append(
                (
                    np.ascontiguousarray(o.idx_gap, dtype=np.int32),
                    np.ascontiguousarray(o.idx_ind, dtype=np.int32),
                    np.ascontiguousarray(o.log_dilution, dtype=np.float64),
                    np.ascontiguousarray(o.od, dtype=np.float64),
                )
            )

    def logp_dlogp(self, theta, i_raw, waner, nthreads=0, want_i=False):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        i_raw = np.ascontiguousarray(i_raw, dtype=np.int8)
        waner = np.ascontiguousarray(waner, dtype=np.int8)
        assert i_raw.shape == (self.G, self.N) and waner.shape == (self.N,)
        lp = C.c_double()
        g = np.empty(17)
        i_out = np.empty((self.G, self.N), dtype=np.int8) if want_i else None
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        (sg, si, sx, sy), (ng, ni, nx, ny) = self.obs
        rc = self.lib.abd_oracle_logp_dlogp(
            C.c_int(self.G), C.c_int(self.N), C.c_int(self.splits.size), p(self.splits), p(self.vacs), p(self.pcr),
            C.c_int64(sg.size), p(sg), p(si), p(sx), p(sy),
            C.c_int64(ng.size), p(ng), p(ni), p(nx), p(ny),
            p(i_raw), p(waner), p(theta), C.byref(lp), p(g), p(i_out), C.c_int(nthreads),
        )
        if rc != 0:
            raise ValueError("abd_oracle_logp_dlogp: bad sizes")
        if want_i:
            return lp.value, g, i_out
        return lp.value, g

    def gibbs_sweep(self, theta, i_raw, waner, chain, seed, sweep, nthreads=0):
        """CPU restatement of abd_gibbs_kernel; returns (i_raw, waner, accepted, proposed) -- inputs are not modified."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        i_raw = np.array(i_raw, dtype=np.int8, order="C", copy=True)
        waner = np.array(waner, dtype=np.int8, order="C", copy=True)
        acc, prop = C.c_int64(), C.c_int64()
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        (sg, si, sx, sy), (ng, ni, nx, ny) = self.obs
        rc = self.lib.abd_oracle_gibbs_sweep(
            C.c_int(self.G), C.c_int(self.N), C.c_int(self.splits.size), p(self.splits), p(self.vacs), p(self.pcr),
            C.c_int64(sg.size), p(sg), p(si), p(sx), p(sy),
            C.c_int64(ng.size), p(ng), p(ni), p(nx), p(ny),
            p(i_raw), p(waner), p(theta), C.c_int(chain), C.c_uint64(seed), C.c_uint32(sweep),
            C.byref(acc), C.byref(prop), C.c_int(nthreads),
        )
        if rc != 0:
            raise ValueError("abd_oracle_gibbs_sweep: bad sizes")
        return i_raw, waner, acc.value, prop.value

    def philox(self, c0, c1, c2, c3, k0, k1):
        out = (C.c_uint32 * 4)()
        self.lib.abd_oracle_philox(C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), C.c_uint32(k0), C.c_uint32(k1), out)
        return [int(v) for v in out]
