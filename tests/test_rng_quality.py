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


# ---- long streams: a depth-50 glass path draws hundreds of values from one stream --------------------------------------------
@pytest.fixture(scope="module")
def long_streams():
    pix = np.repeat(np.arange(512, dtype=np.uint64) + 77000, 128)          # 512 pixels x 128 samples = 65536 streams
    smp = np.tile(np.arange(128, dtype=np.uint64), 512)
    st, inc = _seed(1, pix, smp)
    return _draws(st, inc, 402)                                            # 402 draws = 134 consecutive triples per stream


def test_long_streams_stay_uniform_at_every_depth(long_streams):
    u = long_streams
    n = u.shape[1]
    for i in list(range(24, 402, 21)) + [399, 400, 401]:                   # 64 bins, 63 dof: 1 - 1e-6 quantile 140
        assert _chi2(np.bincount((u[i] * 64).astype(int), minlength=64), n / 64) < 140.0, i
    # pooled over all positions, fine bins: 4096 bins, 4095 dof: mean 4095, sd 90.5 -> 1 - 1e-6 at ~4540
    pooled = np.bincount((u * 4096).astype(int).ravel(), minlength=4096)
    assert _chi2(pooled, u.size / 4096) < 4540.0
    # a stream's own mean over its 402 draws: sigma = 1 / sqrt(12 * 402); the worst of 65536 streams stays inside 5.5 sigma
    assert np.abs(u.mean(axis=0) - 0.5).max() < 5.5 / np.sqrt(12.0 * 402)


def test_fine_triples_pooled_over_draw_positions(long_streams):
    """Consecutive triples in 32 x 32 x 32 cells (the rejection sampler's candidates), pooled over the 134 triple positions of 65536 streams:
    8.8 M triples, 268 per cell, 32767 dof (mean 32767, sd 256: 1 - 1e-6 quantile ~33990).  A lattice structure of the LCG's triples coarser
    than 1/32 of the cube would show here; every stream has its own increment, i.e. its own shift of the lattice."""
    u = long_streams
    t = (u.reshape(134, 3, -1) * 32).astype(np.int64)
    cell = (t[:, 0] * 1024 + t[:, 1] * 32 + t[:, 2]).ravel()
    counts = np.bincount(cell, minlength=32768)
    assert _chi2(counts, cell.size / 32768) < 33990.0
    # ... and the same for triples that straddle the sampler's phase (draws 1-3, 4-6, ... of a stream that first drew one value)
    t = (u[1:400].reshape(133, 3, -1) * 32).astype(np.int64)
    cell = (t[:, 0] * 1024 + t[:, 1] * 32 + t[:, 2]).ravel()
    assert _chi2(np.bincount(cell, minlength=32768), cell.size / 32768) < 33990.0
    # acceptance rate of the unit-ball test at every triple position: pi / 6
    p = u.reshape(134, 3, -1) * 2.0 - 1.0
    acc = ((p ** 2).sum(axis=1) <= 1.0).mean(axis=1)
    assert np.abs(acc - np.pi / 6.0).max() < 5.0 * np.sqrt(0.25 / u.shape[1])


def test_no_serial_correlation_deep_in_a_stream(long_streams):
    u = long_streams
    bound = 5.5 / np.sqrt(u.shape[1])
    for i in (30, 101, 200, 333):
        for lag in (1, 2, 3, 7, 64):
            assert abs(np.corrcoef(u[i], u[i + lag])[0, 1]) < bound, (i, lag)
