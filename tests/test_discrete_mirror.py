"""DiscreteMirror (abdpymc_amd/_native.py): which uploads a caller's sequence of discrete states turns into.  No GPU:
a stand-in context records the calls and keeps the "device" state, the mirror's logic is what is tested."""
import numpy as np

from abdpymc_amd._native import DiscreteMirror


class FakeCtx:
    def __init__(self, G, N):
        self.G, self.N = G, N
        self.i = np.zeros((G, N), dtype=np.int8)
        self.w = np.zeros(N, dtype=np.int8)
        self.gen = 0
        self.calls = []

    def generation(self, chain):
        return self.gen

    def set_discrete(self, chain, i_raw, waner):
        self.i, self.w = np.array(i_raw, dtype=np.int8), np.array(waner, dtype=np.int8)
        self.gen += 1
        self.calls.append("set")

    def flip_discrete(self, chain, flat):
        if flat < self.G * self.N:
            self.i.ravel()[flat] ^= 1
        else:
            self.w[flat - self.G * self.N] ^= 1
        self.gen += 1
        self.calls.append("flip")

    def device_sweep(self, rng):  # something else rewrites the slot
        self.i = (rng.random(self.i.shape) < 0.3).astype(np.int8)
        self.gen += 1


def test_mirror_keeps_the_device_state_equal_to_the_last_point():
    rng = np.random.default_rng(0)
    ctx = FakeCtx(7, 5)
    a, b = DiscreteMirror(ctx, 0), DiscreteMirror(ctx, 0)  # two callables on ONE chain slot
    i = (rng.random((7, 5)) < 0.2).astype(np.int64)
    w = (rng.random(5) < 0.5).astype(np.int64)
    for step in range(200):
        who = a if rng.random() < 0.5 else b
        kind = rng.integers(5)
        if kind == 0:
            i = i.copy()
            i.ravel()[rng.integers(i.size)] ^= 1
        elif kind == 1:
            w = w.copy()
            w[rng.integers(w.size)] ^= 1
        elif kind == 2:
            i = (rng.random((7, 5)) < 0.2).astype(np.int64)  # many bits at once
        elif kind == 3:
            ctx.device_sweep(rng)
        who.update(i, w)
        np.testing.assert_array_equal(ctx.i, i)
        np.testing.assert_array_equal(ctx.w, w)
    assert "flip" in ctx.calls and "set" in ctx.calls


def test_mirror_counts():
    ctx = FakeCtx(4, 3)
    m = DiscreteMirror(ctx, 0)
    i, w = np.zeros((4, 3), dtype=np.int8), np.ones(3, dtype=np.int8)
    m.update(i, w)
    m.update(i, w)
    m.update(i.copy(), w.copy())
    assert (m.uploads, m.flips, m.hits) == (1, 0, 2)
    i2 = i.copy()
    i2[1, 1] = 1
    m.update(i2, w)
    assert (m.uploads, m.flips) == (1, 1)
    ctx.gen += 1  # someone else wrote the slot
    m.update(i2, w)
    assert m.uploads == 2
    i3 = 1 - i2  # more than MAX_FLIPS bits
    m.update(i3, w)
    assert m.uploads == 3
    m.invalidate()
    m.update(i3, w)
    assert m.uploads == 4
