"""Statistical checks of the sample stream AS THE RENDERER USES IT: a million short streams, one per (pixel, sample) of a block of the
image, seeded by the hash chain of oracle/rtw_oracle.c (rng_seed), a few dozen draws each -- not one long sequence.

The generator is a 32-bit LCG with a per-stream odd increment whose draws are the top 24 bits of the state (the permutation stage of
PCG RXS-M-XS that rounds 1-2a ran on top cost 9 % of the GPU frame, DESIGN.md "RNG").  What a path tracer needs of it: uniform draws
at every position of the stream, uniform consecutive triples (the unit-vector rejection sampler of vec3.rs:228-239 consumes them),
no correlation along a stream, and none between the streams of neighbouring samples and pixels.  Thresholds are chi-square quantiles
around p = 1e-6 for the degrees of freedom stated, so a sound generator fails a run about once in a million; the numpy model is pinned to the C
oracle first (which the GPU matches bit for bit, tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_binding as O

M = np.uint64(0xFFFFFFFF)


def _mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    return x


def _seed(seed, pixel, sample):
    h = _mix(np.full(pixel.shape, (seed + 0x9E3779B9) & 0xFFFFFFFF, dtype=np.uint64))
    h = _mix(h ^ np.uint64(seed >> 32)); h = _mix(h ^ pixel.astype(np.uint64)); h = _mix(h ^ sample.astype(np.uint64))
    return h, _mix(h ^ np.uint64(0x85EBCA6B)) | np.uint64(1)


def _draws(state, inc, n):
    out = np.empty((n,) + state.shape, dtype=np.float64)
    s = state.copy()
    for i in range(n):
        out[i] = (s >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
        s = (s * np.uint64(747796405) + inc) & M
    return out


def _chi2(counts, expected):
    return float(((counts - expected) ** 2 / expected).sum())


@pytest.fixture(scope="module")
def streams():
    pix = np.repeat(np.arange(4096, dtype=np.uint64) + 1000, 256)         # 4096 pixels x 256 samples = 1 M streams
    smp = np.tile(np.arange(256, dtype=np.uint64), 4096)
    st, inc = _seed(1, pix, smp)
    return st, inc, _draws(st, inc, 24)


def test_numpy_model_is_the_oracles_stream():
    for seed, pixel, sample in ((1, 1000, 0), (1, 5095, 255), (0xDEADBEEFCAFEF00D, 89999, 499)):
        st = (C.c_uint32 * 2)()
        O.lib().rtw_oracle_rng_seed(seed, pixel, sample, st)
        h, inc = _seed(seed, np.array([pixel], dtype=np.uint64), np.array([sample], dtype=np.uint64))
        assert (st[0], st[1]) == (int(h[0]), int(inc[0]))
        want = _draws(h, inc, 40)[:, 0]
        assert [O.lib().rtw_oracle_rng_next(st) for _ in range(40)] == [float(np.float32(x)) for x in want]


def test_every_draw_position_is_uniform(streams):
    _, _, u = streams
    n = u.shape[1]
    for i in range(24):                                                    # 64 bins, 63 dof: chi2 quantile at 1 - 1e-6 is 140
        assert _chi2(np.bincount((u[i] * 64).astype(int), minlength=64), n / 64) < 140.0, i
    lo = (u[3] * 2 ** 24).astype(np.int64) & 255                           # the low byte of a draw (state bits 8-15), 255 dof: 380
    assert _chi2(np.bincount(lo, minlength=256), n / 256) < 380.0
    assert abs(u.mean() - 0.5) < 5.0 / np.sqrt(12.0 * u.size)


def test_consecutive_triples_fill_the_cube_and_the_sphere(streams):
    _, _, u = streams
    t = u.reshape(8, 3, -1)
    n = t.shape[2]
    for k in range(8):                                                     # 8 x 8 x 8 cells, 511 dof: chi2 quantile at 1 - 1e-6 is 680
        cell = (t[k, 0] * 8).astype(int) * 64 + (t[k, 1] * 8).astype(int) * 8 + (t[k, 2] * 8).astype(int)
        assert _chi2(np.bincount(cell, minlength=512), n / 512) < 680.0, k
    # random_unit_vec (vec3.rs:228-239): first triple of [-1, 1)^3 inside the unit ball, normalised
    p = t * 2.0 - 1.0
    acc = (p ** 2).sum(axis=1) <= 1.0
    assert abs(acc[0].mean() - np.pi / 6.0) < 5.0 * np.sqrt(0.25 / n)
    first = np.argmax(acc, axis=0)
    sel = p[first, :, np.arange(n)][acc.any(axis=0)]
    sel /= np.sqrt((sel ** 2).sum(axis=1))[:, None]
    for axis in range(3):                                                  # a uniform direction has uniform coordinates (Archimedes)
        assert _chi2(np.bincount(((sel[:, axis] + 1.0) * 32).astype(int).clip(0, 63), minlength=64), len(sel) / 64) < 140.0, axis
    octant = (sel[:, 0] > 0) * 4 + (sel[:, 1] > 0) * 2 + (sel[:, 2] > 0)
    assert _chi2(np.bincount(octant, minlength=8), len(sel) / 8) < 45.0    # 7 dof: 1 - 1e-6 quantile is 40


def test_no_correlation_along_or_across_streams(streams):
    _, _, u = streams
    n = u.shape[1]
    bound = 5.5 / np.sqrt(n)                                               # correlation of independent uniforms: sigma = 1 / sqrt(n)
    for lag in (1, 2, 3, 6):
        for i in (0, 1, 4, 9, 17):
            assert abs(np.corrcoef(u[i], u[i + lag])[0, 1]) < bound, (i, lag)
    grid = u.reshape(24, 4096, 256)
    for i in (0, 1, 2, 5, 11):
        assert abs(np.corrcoef(grid[i, :, :-1].ravel(), grid[i, :, 1:].ravel())[0, 1]) < bound      # neighbouring samples of a pixel
        assert abs(np.corrcoef(grid[i, :-1, :].ravel(), grid[i, 1:, :].ravel())[0, 1]) < bound      # the same sample of neighbouring pixels
        assert abs(np.corrcoef(grid[i, :, :-1].ravel(), grid[i + 1, :, 1:].ravel())[0, 1]) < bound  # draw i of sample s, draw i + 1 of s + 1
