"""
ctypes binding of ``libabd_hip.so`` (C ABI: ``include/abd_hip.h``).

The HIP library is the only implementation of the hot path in this package: if it is missing or fails
to load, importing the symbols below raises -- there is no CPU fallback (the CPU restatement under
``oracle/`` is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import numpy as np

N_THETA = 17
MAX_GAPS = 512
STORE_F64, STORE_F32 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABD_HIP_LIB", os.path.join(_HERE, "libabd_hip.so"))


class AbdError(RuntimeError):
    """HIP/runtime failure inside the native library."""


class _AntigenObs(C.Structure):
    _fields_ = [
        ("n_obs", C.c_int64),
        ("idx_gap", C.POINTER(C.c_int32)),
        ("idx_ind", C.POINTER(C.c_int32)),
        ("log_dilution", C.POINTER(C.c_double)),
        ("od", C.POINTER(C.c_double)),
    ]


class _Desc(C.Structure):
    _fields_ = [
        ("n_gaps", C.c_int32),
        ("n_inds", C.c_int32),
        ("n_splits", C.c_int32),
        ("splits", C.c_int32 * 2),
        ("storage", C.c_int32),
        ("n_chain_slots", C.c_int32),
        ("device", C.c_int32),
        ("s", _AntigenObs),
        ("n", _AntigenObs),
        ("vacs", C.POINTER(C.c_int8)),
        ("pcrpos", C.POINTER(C.c_int8)),
    ]


class _SamplerOpts(C.Structure):
    _fields_ = [
        ("tune", C.c_int64),
        ("seed", C.c_uint64),
        ("target_accept", C.c_double),
        ("max_treedepth", C.c_int32),
        ("gibbs", C.c_int32),
        ("accumulate", C.c_int32),
        ("chain_offset", C.c_int32),
        ("dense_metric", C.c_int32),
        ("reserved", C.c_int32),
    ]


class _Record(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("first", C.c_int64),
        ("thin", C.c_int64),
        ("i_raw", C.POINTER(C.c_int8)),
        ("ab_s_waner", C.POINTER(C.c_int8)),
        ("i", C.POINTER(C.c_int8)),
        ("ab_n_mu", C.POINTER(C.c_double)),
        ("ab_s_mu", C.POINTER(C.c_double)),
    ]


N_STATS = 11
STAT_NAMES = ("lp", "tree_depth", "n_steps", "mean_tree_accept", "step_size", "diverging", "energy", "max_energy_error",
              "gibbs_accepted", "gibbs_proposed", "t_done")

_lib = None

# every symbol include/abd_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
# array arguments (double*, int32_t*, int8_t*) are passed as plain addresses: building a typed ctypes pointer per argument
# (ndarray.ctypes.data_as) costs 1.5-3 us each, which is a third of a synchronous call on a small cohort
_D = C.c_void_p
_I32 = C.c_void_p
_I8 = C.c_void_p
SYMBOLS = {
    "abd_version": (C.c_char_p, []),
    "abd_last_error": (C.c_char_p, []),
    "abd_create": (C.c_int, [C.POINTER(_Desc), C.POINTER(_P)]),
    "abd_destroy": (C.c_int, [_P]),
    "abd_device_name": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "abd_set_discrete": (C.c_int, [_P, C.c_int32, _I8, _I8]),
    "abd_flip_discrete": (C.c_int, [_P, C.c_int32, C.c_int64]),
    "abd_get_discrete": (C.c_int, [_P, C.c_int32, _I8, _I8]),
    "abd_gibbs_sweep": (C.c_int, [_P, C.c_int32, _I32, _D, C.c_uint64, C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "abd_logp": (C.c_int, [_P, C.c_int32, _D, _D]),
    "abd_logp_dlogp": (C.c_int, [_P, C.c_int32, _D, _D, _D]),
    "abd_loglik_dlogp": (C.c_int, [_P, C.c_int32, _D, _D, _D]),
    "abd_logp_dlogp_batch": (C.c_int, [_P, C.c_int32, _I32, _D, _D, _D]),
    "abd_n_result_slots": (C.c_int, [_P]),
    "abd_logp_dlogp_batch_enqueue": (C.c_int, [_P, C.c_int32, C.c_int32, _I32, _D]),
    "abd_wait": (C.c_int, [_P]),
    "abd_fetch": (C.c_int, [_P, C.c_int32, _D, _D]),
    "abd_fetch_many": (C.c_int, [_P, C.c_int32, _I32, _D, _D]),
    "abd_logp_dlogp_many": (C.c_int, [_P, C.c_int32, C.c_int32, _I32, _D, _D, _D]),
    "abd_deterministics": (C.c_int, [_P, C.c_int32, _D, _I8, _D, _D]),
    "abd_sampler_create": (C.c_int, [_P, C.c_int32, _I32, _D, C.POINTER(_SamplerOpts), C.POINTER(_P)]),
    "abd_sampler_destroy": (None, [_P]),
    "abd_sampler_run": (C.c_int, [_P, C.c_int64, _D, _D]),
    "abd_sampler_run_record": (C.c_int, [_P, C.c_int64, _D, _D, C.POINTER(_Record)]),
    "abd_sampler_set_adaptation": (C.c_int, [_P, C.c_int32, _D, C.c_double]),
    "abd_sampler_means": (C.c_int, [_P, C.c_int32, _D, _D, _D, C.POINTER(C.c_int64)]),
    "abd_sampler_adaptation": (C.c_int, [_P, C.c_int32, _D, _D, _D]),
    "abd_theta_prior": (C.c_int, [_P, _D, _D, _D]),
    "abd_set_individual_offset": (C.c_int, [_P, C.c_int64]),
    "abd_kernel_timing": (C.c_int, [_P, C.c_int32]),
    "abd_kernel_time": (C.c_int, [_P, _D, C.POINTER(C.c_int64), C.c_int32]),
    "abd_set_launch_config": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "abd_algorithmic_bytes": (C.c_int64, [_P, C.c_int32]),
    "abd_wait_fallbacks": (C.c_int64, [_P]),
    "abd_stream_queues": (C.c_int, [_P, _I32, C.c_int32]),
    "abd_n_pipes": (C.c_int, [_P]),
    "abd_is_dense": (C.c_int, [_P]),
}


def load():
    """Load the HIP library (once).  Raises ImportError loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"abdpymc_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the hot path."
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _err(lib) -> str:
    return lib.abd_last_error().decode("utf-8", "replace")


def _check(lib, rc: int):
    if rc == 0:
        return
    msg = _err(lib)
    if rc == -1:  # ABD_ERR_ARG -> the reference raises ValueError for the same conditions
        raise ValueError(msg)
    raise AbdError(f"[{rc}] {msg}")


def _as(arr, dtype):
    return np.ascontiguousarray(arr, dtype=dtype)


def _tptr(arr, ctype):
    """Typed ctypes pointer to an array's data (structure fields, typed out-parameters)."""
    return arr.ctypes.data_as(C.POINTER(ctype))


def _ptr(arr, ctype=None):
    """Address of a C-contiguous array's data (the array must stay referenced by the caller for the duration of the call).
    INPUT arguments only (the library reads through it): a read-only array is fine here."""
    try:
        return C.addressof(C.c_char.from_buffer(arr))  # 0.4 us; needs a writable, non-empty buffer
    except (TypeError, ValueError, BufferError):
        return arr.ctypes.data


def _out(arr, dtype, shape=None, name="output array"):
    """Address of an array the LIBRARY WRITES: it must be writeable, C-contiguous, of the exact dtype (ctypes sees only an
    address, so nothing else would catch a read-only memmap or a float32 buffer before the library writes through it)."""
    if not isinstance(arr, np.ndarray) or arr.dtype != dtype or not arr.flags.c_contiguous or not arr.flags.writeable:
        raise ValueError(f"{name} must be a writeable C-contiguous {np.dtype(dtype).name} ndarray")
    if shape is not None and arr.shape != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {arr.shape}")
    return _ptr(arr)


class Context:
    """
    Device-resident cohort + chain slots.  Mirrors what ``abd.model(data, splits, ignore_pcrpos)``
    closes over (reference abd.py:396-442).
    """

    def __init__(
        self,
        n_gaps: int,
        n_inds: int,
        s_obs,  # (idx_gap, idx_ind, log_dilution, od)
        n_obs,
        vacs,  # (N, G)
        pcrpos,  # (N, G) or None (= ignore_pcrpos)
        splits: Optional[Sequence[int]] = None,
        n_chains: int = 1,
        storage: str = "f64",
        device: int = -1,
    ):
        lib = load()
        self._lib = lib
        self._h = _P()
        self._samplers = weakref.WeakSet()  # destroyed before the context they point into
        self.n_gaps, self.n_inds, self.n_chains = int(n_gaps), int(n_inds), int(n_chains)
        d = _Desc()
        d.n_gaps, d.n_inds = self.n_gaps, self.n_inds
        splits = tuple(splits or ())
        if len(splits) > 2:
            raise NotImplementedError("only implemented 1-3 time chunks (0-2 splits)")
        d.n_splits = len(splits)
        for k, s in enumerate(splits):
            d.splits[k] = int(s)
        d.storage = {"f64": STORE_F64, "f32": STORE_F32}[storage]
        d.n_chain_slots = self.n_chains
        d.device = device
        keep = []

        def obs(o):
            g, j, x, y = o
            g, j = _as(g, np.int32), _as(j, np.int32)
            x, y = _as(x, np.float64), _as(y, np.float64)
            if not (g.shape == j.shape == x.shape == y.shape and g.ndim == 1):
                raise ValueError("observation arrays must be 1-D and of equal length")
            keep.extend([g, j, x, y])
            a = _AntigenObs()
            a.n_obs = g.size
            a.idx_gap, a.idx_ind = _tptr(g, C.c_int32), _tptr(j, C.c_int32)
            a.log_dilution, a.od = _tptr(x, C.c_double), _tptr(y, C.c_double)
            return a

        d.s, d.n = obs(s_obs), obs(n_obs)
        vacs = np.asarray(vacs)
        if vacs.shape != (self.n_inds, self.n_gaps):
            raise ValueError(f"vacs shape {vacs.shape} != (n_inds, n_gaps) = {(self.n_inds, self.n_gaps)}")
        v8 = _as(vacs, np.int8)
        if not np.array_equal(v8, vacs):
            raise ValueError("vacs must be 0/1")
        keep.append(v8)
        d.vacs = _tptr(v8, C.c_int8)
        if pcrpos is not None:
            pcrpos = np.asarray(pcrpos)
            if vacs.shape != pcrpos.shape:
                raise ValueError("vacs and pcrpos are different shapes")  # abd.py:196-197
            p8 = _as(pcrpos, np.int8)
            if not np.array_equal(p8, pcrpos):
                raise ValueError("pcrpos must be 0/1")
            keep.append(p8)
            d.pcrpos = _tptr(p8, C.c_int8)
        _check(lib, lib.abd_create(C.byref(d), C.byref(self._h)))
        self.n_result_slots = lib.abd_n_result_slots(self._h)
        # one counter per chain slot, bumped by everything that rewrites the slot's device-side discrete state
        # (set_discrete, flip_discrete, gibbs_sweep, the native sampler): host mirrors of that state (DiscreteMirror)
        # are only trusted while the counter stands where they left it
        self._generation = [0] * self.n_chains

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for smp in list(self._samplers):
                smp.close()
            self._lib.abd_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- info -------------------------------------------------------------------------------
    @property
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(self._lib, self._lib.abd_device_name(self._h, buf, 256))
        return buf.value.decode()

    @property
    def is_dense(self) -> bool:
        return bool(self._lib.abd_is_dense(self._h))

    @property
    def n_pipes(self) -> int:
        """HIP streams that stream-ordered dense launches rotate over."""
        return int(self._lib.abd_n_pipes(self._h))

    @property
    def wait_fallbacks(self) -> int:
        """Synchronous calls that had to fall back from the polled completion tag to a stream synchronise (expect 0)."""
        return int(self._lib.abd_wait_fallbacks(self._h))

    def stream_queues(self):
        """Hardware queue of each of the context's 8 HIP streams (streams with the same number serialise their kernels)."""
        q = (C.c_int32 * 8)()
        _check(self._lib, self._lib.abd_stream_queues(self._h, q, 8))
        return list(q)

    def algorithmic_bytes(self, n_chains: int) -> int:
        return int(self._lib.abd_algorithmic_bytes(self._h, n_chains))

    # -- discrete state -----------------------------------------------------------------------
    def generation(self, chain: int) -> int:
        """Changes every time the slot's device-side (i_raw, ab_s_waner) is rewritten by anyone."""
        return self._generation[int(chain)]

    def _bump(self, chains):
        for c in np.atleast_1d(chains):
            self._generation[int(c)] += 1

    def set_discrete(self, chain: int, i_raw, waner):
        i_raw, waner = np.asarray(i_raw), np.asarray(waner)
        if i_raw.shape != (self.n_gaps, self.n_inds):
            raise ValueError(f"i_raw shape {i_raw.shape} != (n_gaps, n_inds) = {(self.n_gaps, self.n_inds)}")
        if waner.shape != (self.n_inds,):
            raise ValueError(f"ab_s_waner shape {waner.shape} != ({self.n_inds},)")
        i8, w8 = _as(i_raw, np.int8), _as(waner, np.int8)
        if not (np.array_equal(i8, i_raw) and np.array_equal(w8, waner)):
            raise ValueError("i_raw / ab_s_waner must be 0/1")
        _check(self._lib, self._lib.abd_set_discrete(self._h, chain, _ptr(i8, C.c_int8), _ptr(w8, C.c_int8)))
        self._bump(chain)

    def flip_discrete(self, chain: int, flat: int):
        _check(self._lib, self._lib.abd_flip_discrete(self._h, chain, int(flat)))
        self._bump(chain)

    def get_discrete(self, chain: int):
        i = np.empty((self.n_gaps, self.n_inds), dtype=np.int8)
        w = np.empty(self.n_inds, dtype=np.int8)
        _check(self._lib, self._lib.abd_get_discrete(self._h, chain, _ptr(i, C.c_int8), _ptr(w, C.c_int8)))
        return i, w

    def gibbs_sweep(self, chains, theta, seed: int, sweep: int):
        """One binary Gibbs-Metropolis sweep of each chain's [i_raw, ab_s_waner] in place -> (accepted, proposed)."""
        ch = _as(np.atleast_1d(chains), np.int32)
        t = _as(np.atleast_2d(theta), np.float64)
        if t.shape != (ch.size, N_THETA):
            raise ValueError(f"theta must have shape ({ch.size}, {N_THETA})")
        acc = np.zeros(ch.size, dtype=np.int64)
        prop = np.zeros(ch.size, dtype=np.int64)
        self._bump(ch)
        _check(self._lib, self._lib.abd_gibbs_sweep(self._h, ch.size, _ptr(ch, C.c_int32), _ptr(t, C.c_double),
                                                    C.c_uint64(seed & (2**64 - 1)), C.c_uint32(sweep & 0xFFFFFFFF),
                                                    _tptr(acc, C.c_int64), _tptr(prop, C.c_int64)))
        return acc, prop

    # -- evaluations ------------------------------------------------------------------------
    def logp(self, chain: int, theta) -> float:
        t = _as(theta, np.float64)
        if t.shape != (N_THETA,):
            raise ValueError(f"theta must have shape ({N_THETA},)")
        out = C.c_double()
        _check(self._lib, self._lib.abd_logp(self._h, chain, _ptr(t, C.c_double), C.byref(out)))
        return out.value

    def logp_dlogp(self, chain: int, theta):
        t = _as(theta, np.float64)
        if t.shape != (N_THETA,):
            raise ValueError(f"theta must have shape ({N_THETA},)")
        out = C.c_double()
        g = np.empty(N_THETA)
        _check(self._lib, self._lib.abd_logp_dlogp(self._h, chain, _ptr(t, C.c_double), C.byref(out), _ptr(g, C.c_double)))
        return out.value, g

    def loglik_dlogp(self, chain: int, theta):
        """Data term only (the two observed Normals) and its gradient w.r.t. theta."""
        t = _as(theta, np.float64)
        if t.shape != (N_THETA,):
            raise ValueError(f"theta must have shape ({N_THETA},)")
        out = C.c_double()
        g = np.empty(N_THETA)
        _check(self._lib, self._lib.abd_loglik_dlogp(self._h, chain, _ptr(t, C.c_double), C.byref(out), _ptr(g, C.c_double)))
        return out.value, g

    def logp_dlogp_batch(self, chains, theta):
        ch = _as(chains, np.int32)
        t = _as(theta, np.float64)
        if t.shape != (ch.size, N_THETA):
            raise ValueError(f"theta must have shape ({ch.size}, {N_THETA})")
        lp = np.empty(ch.size)
        g = np.empty((ch.size, N_THETA))
        _check(
            self._lib,
            self._lib.abd_logp_dlogp_batch(
                self._h, ch.size, _ptr(ch, C.c_int32), _ptr(t, C.c_double), _ptr(lp, C.c_double), _ptr(g, C.c_double)
            ),
        )
        return lp, g

    def enqueue(self, slot: int, chains, theta):
        ch = _as(chains, np.int32)
        t = _as(theta, np.float64)
        if t.shape != (ch.size, N_THETA):
            raise ValueError(f"theta must have shape ({ch.size}, {N_THETA})")
        _check(
            self._lib,
            self._lib.abd_logp_dlogp_batch_enqueue(self._h, slot, ch.size, _ptr(ch, C.c_int32), _ptr(t, C.c_double)),
        )

    def wait(self):
        _check(self._lib, self._lib.abd_wait(self._h))

    def fetch(self, slot: int, n: int):
        lp = np.empty(n)
        g = np.empty((n, N_THETA))
        _check(self._lib, self._lib.abd_fetch(self._h, slot, _ptr(lp, C.c_double), _ptr(g, C.c_double)))
        return lp, g

    def fetch_many(self, slots, n_per_slot: int, out_lp=None, out_g=None):
        """fetch() for several slots at once -> (len(slots), n_per_slot) logp and (len(slots), n_per_slot, 17) grad;
        ``out_lp`` / ``out_g``: C-contiguous float64 arrays of those shapes to write into instead of new ones."""
        sl = _as(slots, np.int32)
        lp = np.empty((sl.size, n_per_slot)) if out_lp is None else out_lp
        g = np.empty((sl.size, n_per_slot, N_THETA)) if out_g is None else out_g
        _check(self._lib, self._lib.abd_fetch_many(self._h, sl.size, _ptr(sl, C.c_int32),
                                                   _out(lp, np.float64, (sl.size, n_per_slot), "out_lp"),
                                                   _out(g, np.float64, (sl.size, n_per_slot, N_THETA), "out_g")))
        return lp, g

    def logp_dlogp_many(self, chains, thetas, out_lp=None, out_g=None):
        """``thetas`` (K, n, 17): K independent evaluations of the same n chains, stream-ordered inside the library
        (enqueue all, wait once, fetch) -> logp (K, n), grad (K, n, 17); ``out_lp`` / ``out_g`` to write into."""
        ch = _as(chains, np.int32)
        t = _as(thetas, np.float64)
        if t.ndim != 3 or t.shape[1:] != (ch.size, N_THETA):
            raise ValueError(f"thetas must have shape (K, {ch.size}, {N_THETA})")
        K = t.shape[0]
        lp = np.empty((K, ch.size)) if out_lp is None else out_lp
        g = np.empty((K, ch.size, N_THETA)) if out_g is None else out_g
        _check(self._lib, self._lib.abd_logp_dlogp_many(self._h, K, ch.size, _ptr(ch, C.c_int32), _ptr(t, C.c_double),
                                                        _out(lp, np.float64, (K, ch.size), "out_lp"),
                                                        _out(g, np.float64, (K, ch.size, N_THETA), "out_g")))
        return lp, g

    def deterministics(self, chain: int, theta):
        t = _as(theta, np.float64)
        G, N = self.n_gaps, self.n_inds
        i = np.empty((G, N), dtype=np.int8)
        mun = np.empty((G, N))
        mus = np.empty((G, N))
        _check(
            self._lib,
            self._lib.abd_deterministics(
                self._h, chain, _ptr(t, C.c_double), _ptr(i, C.c_int8), _ptr(mun, C.c_double), _ptr(mus, C.c_double)
            ),
        )
        return i, mun, mus

    def theta_prior(self, theta):
        """The theta-only part of the joint logp (continuous priors + Jacobians) and its gradient."""
        t = _as(theta, np.float64)
        if t.shape != (N_THETA,):
            raise ValueError(f"theta must have shape ({N_THETA},)")
        out = C.c_double()
        g = np.empty(N_THETA)
        _check(self._lib, self._lib.abd_theta_prior(self._h, _ptr(t, C.c_double), C.byref(out), _ptr(g, C.c_double)))
        return out.value, g

    def set_individual_offset(self, first_individual: int):
        _check(self._lib, self._lib.abd_set_individual_offset(self._h, int(first_individual)))

    def sampler(self, chains, theta0, tune: int, seed: int = 0, target_accept: float = 0.8, max_treedepth: int = 10,
                gibbs: bool = True, accumulate: bool = False, chain_offset: int = 0,
                dense_metric: bool = False) -> "NativeSampler":
        """The compound step [NUTS; Gibbs sweep] for several chains, driven inside the library: the chains advance as
        independent units on their own HIP streams (abd_hip.h: abd_sampler_create)."""
        return NativeSampler(self, chains, theta0, tune, seed, target_accept, max_treedepth, gibbs, accumulate, chain_offset,
                             dense_metric)

    # -- measurement --------------------------------------------------------------------------
    def kernel_timing(self, mode):
        """0/False off; 1/True HIP events per launch, launches serialised (isolated kernel); 2 per window of
        stream-ordered launches in their real launch shape."""
        _check(self._lib, self._lib.abd_kernel_timing(self._h, int(mode)))

    def set_launch_config(self, blocks: int = 0, chains_per_wave: int = 0):
        _check(self._lib, self._lib.abd_set_launch_config(self._h, int(blocks), int(chains_per_wave)))

    def kernel_time(self, reset: bool = True):
        ms = C.c_double()
        n = C.c_int64()
        _check(self._lib, self._lib.abd_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


class DiscreteMirror:
    """
    Host mirror of ONE chain slot's device-side ``(i_raw, ab_s_waner)``, for callers that are handed the whole
    discrete state on every call (``Model.compile_logp()``'s point function under BinaryGibbsMetropolis, a PyTensor
    ``Op.perform``): ``update`` uploads only what changed -- nothing, a few ``abd_flip_discrete`` calls, or the whole
    panel.  The mirror is trusted only while the slot's generation counter (``Context.generation``) stands where
    this mirror left it: anything else that rewrites the slot (another callable's ``set_discrete``, a device Gibbs
    sweep, the native sampler) bumps the counter and the next ``update`` re-uploads.
    """

    MAX_FLIPS = 8

    def __init__(self, ctx: "Context", chain: int = 0):
        self.ctx, self.chain = ctx, int(chain)
        self._state = None
        self._gen = None
        self.uploads = self.flips = self.hits = 0  # what update() did so far (tests, profiling)

    def invalidate(self):
        self._state = None

    def adopt(self, i_raw, waner):
        """The device holds exactly this state now (e.g. it was just read back after a device sweep): remember it under
        the slot's current generation without uploading anything."""
        i_raw, waner = np.asarray(i_raw), np.asarray(waner)
        self._state = (i_raw.copy(), waner.copy(), i_raw, waner)
        self._gen = self.ctx.generation(self.chain)

    def update(self, i_raw, waner):
        ctx, c = self.ctx, self.chain
        i_raw, waner = np.asarray(i_raw), np.asarray(waner)
        prev = self._state
        if prev is not None and self._gen == ctx.generation(c) and prev[0].shape == i_raw.shape and prev[1].shape == waner.shape:
            if i_raw is prev[2] and waner is prev[3] and not (i_raw.flags.writeable or waner.flags.writeable):
                self.hits += 1  # the very same read-only arrays: nothing can have changed
                return
            di = np.flatnonzero(i_raw.ravel() != prev[0].ravel())
            dw = np.flatnonzero(waner != prev[1])
            if di.size + dw.size == 0:
                self.hits += 1
                return
            if di.size + dw.size <= self.MAX_FLIPS:
                for f in di:
                    ctx.flip_discrete(c, int(f))
                for f in dw:
                    ctx.flip_discrete(c, int(i_raw.size + f))
                self.flips += int(di.size + dw.size)
                self._state = (i_raw.copy(), waner.copy(), i_raw, waner)
                self._gen = ctx.generation(c)
                return
        ctx.set_discrete(c, i_raw, waner)
        self.uploads += 1
        self._state = (i_raw.copy(), waner.copy(), i_raw, waner)
        self._gen = ctx.generation(c)


class NativeSampler:
    """``abd_sampler_*``: what ``pm.sample`` runs for this model (abd.py:921-922): independent chains, one evaluation launch
    per leapfrog of a unit of 1-4 chains, leapfrog trains (abd_hip.h)."""

    def __init__(self, ctx: Context, chains, theta0, tune, seed, target_accept, max_treedepth, gibbs, accumulate,
                 chain_offset=0, dense_metric=False):
        self._ctx = ctx  # keeps the context alive
        self._lib = ctx._lib
        self._h = _P()
        ch = _as(np.atleast_1d(chains), np.int32)
        t0 = _as(np.atleast_2d(theta0), np.float64)
        if t0.shape != (ch.size, N_THETA):
            raise ValueError(f"theta0 must have shape ({ch.size}, {N_THETA})")
        self.n = int(ch.size)
        self._chains = ch.copy()
        o = _SamplerOpts()
        o.tune, o.seed = int(tune), int(seed) & (2**64 - 1)
        o.target_accept, o.max_treedepth = float(target_accept), int(max_treedepth)
        o.gibbs, o.accumulate = int(bool(gibbs)), int(bool(accumulate))
        o.chain_offset = int(chain_offset)
        o.dense_metric = int(bool(dense_metric))
        _check(self._lib, self._lib.abd_sampler_create(ctx._h, self.n, _ptr(ch, C.c_int32), _ptr(t0, C.c_double), C.byref(o),
                                                       C.byref(self._h)))
        ctx._samplers.add(self)

    def run(self, n_iter: int):
        """Advance all chains by n_iter iterations -> theta (n, n_iter, 17), stats {name: (n, n_iter)}."""
        theta = np.empty((self.n, n_iter, N_THETA))
        stats = np.empty((self.n, n_iter, N_STATS))
        self._ctx._bump(self._chains)  # the sweeps rewrite the chains' discrete state
        _check(self._lib, self._lib.abd_sampler_run(self._h, int(n_iter), _ptr(theta, C.c_double), _ptr(stats, C.c_double)))
        return theta, {name: stats[:, :, k].copy() for k, name in enumerate(STAT_NAMES)}

    def run_record(self, n_iter: int, first: int, i_raw=None, ab_s_waner=None, i=None, ab_n_mu=None, ab_s_mu=None, thin: int = 1):
        """
        run() that also writes every iteration's discrete state / Deterministics of every chain into the given
        C-contiguous arrays of shape (n, capacity, G, N) (int8 for i_raw and i, float64 for the two mu) and
        (n, capacity, N) int8 for ab_s_waner, at draws first .. first + n_iter - 1.  thin = K > 1: only iterations
        0, K, 2K, ... of the call are written, at draws first, first + 1, ... (ceil(n_iter / K) of them).
        """
        if thin < 1:
            raise ValueError(f"thin must be >= 1, got {thin}")
        G, N = self._ctx.n_gaps, self._ctx.n_inds
        rec = _Record()
        cap = None
        for name, arr, dt, tail in (("i_raw", i_raw, np.int8, (G, N)), ("ab_s_waner", ab_s_waner, np.int8, (N,)),
                                    ("i", i, np.int8, (G, N)), ("ab_n_mu", ab_n_mu, np.float64, (G, N)),
                                    ("ab_s_mu", ab_s_mu, np.float64, (G, N))):
            if arr is None:
                continue
            if arr.dtype != dt or not arr.flags.c_contiguous or not arr.flags.writeable or arr.shape[0] != self.n or arr.shape[2:] != tail:
                raise ValueError(f"{name}: need a writeable C-contiguous {np.dtype(dt).name} array of shape (n, capacity) + {tail}")
            if cap is not None and arr.shape[1] != cap:
                raise ValueError("record arrays differ in capacity")
            cap = arr.shape[1]
            setattr(rec, name, _tptr(arr, C.c_int8 if dt == np.int8 else C.c_double))
        rec.capacity, rec.first, rec.thin = int(cap or 0), int(first), int(thin)
        theta = np.empty((self.n, n_iter, N_THETA))
        stats = np.empty((self.n, n_iter, N_STATS))
        self._ctx._bump(self._chains)
        _check(self._lib, self._lib.abd_sampler_run_record(self._h, int(n_iter), _ptr(theta, C.c_double), _ptr(stats, C.c_double),
                                                           C.byref(rec)))
        return theta, {name: stats[:, :, k].copy() for k, name in enumerate(STAT_NAMES)}

    def means(self, k: int):
        """Posterior means of i, ab_n_mu, ab_s_mu of the k-th chain over the accumulated draws, and their count."""
        G, N = self._ctx.n_gaps, self._ctx.n_inds
        out = [np.empty((G, N)) for _ in range(3)]
        n = C.c_int64()
        _check(self._lib, self._lib.abd_sampler_means(self._h, int(k), *[_ptr(o, C.c_double) for o in out], C.byref(n)))
        return out[0], out[1], out[2], n.value

    def adaptation(self, k: int):
        """(diagonal of M^-1, step size) of the k-th chain."""
        inv_mass = np.empty(N_THETA)
        eps = C.c_double()
        _check(self._lib, self._lib.abd_sampler_adaptation(self._h, int(k), _ptr(inv_mass, C.c_double), C.byref(eps), None))
        return inv_mass, eps.value

    def set_adaptation(self, k: int, inv_mass=None, step_size: float = 0.0):
        """Install a diagonal M^-1 and / or a step size for the k-th chain (between two runs)."""
        im = None if inv_mass is None else np.ascontiguousarray(inv_mass, dtype=np.float64)
        if im is not None and im.shape != (N_THETA,):
            raise ValueError(f"inv_mass must have {N_THETA} entries")
        _check(self._lib, self._lib.abd_sampler_set_adaptation(self._h, int(k), None if im is None else _ptr(im, C.c_double), float(step_size)))

    def metric(self, k: int) -> np.ndarray:
        """Full M^-1 (17 x 17) of the k-th chain."""
        m = np.empty((N_THETA, N_THETA))
        _check(self._lib, self._lib.abd_sampler_adaptation(self._h, int(k), None, None, _ptr(m, C.c_double)))
        return m

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.abd_sampler_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
