"""Distribution of the counter-based dropout / noise stream (oracle/rng.py restates csrc/dvs_device.h: dvs_site_key,
dvs_draw; the emulator and GPU train-mode tests check that the device draws exactly these bits).  The stream is the build's
own choice — what has to agree with the reference is its distribution: Bernoulli(1 - p) keep decisions that are
independent across elements, halves of a draw, DAGs and sites, and standard-normal reparameterisation noise."""
import numpy as np

from oracle import rng


def _corr(a, b):
    a = a - a.mean()
    b = b - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


def test_keep_decisions_are_bernoulli_and_uncorrelated():
    thr = int(np.rint(np.float32(0.15) * np.float32(65536.0)))
    dags = np.arange(3000, dtype=np.uint64)
    keys = rng.site_key((7 << 32) | 12345, 3, dags)                   # consecutive DAGs of one site, as a batch has them
    pairs = np.arange(1024, dtype=np.uint64)
    h = rng.draw(keys[:, None], pairs[None, :])
    lo = ((h & np.uint64(0xFFFF)) >= thr).astype(np.float64)
    hi = ((h >> np.uint64(16)) >= thr).astype(np.float64)
    n = lo.size
    for k in (lo, hi):
        assert abs(k.mean() - (1 - thr / 65536.0)) < 4 * np.sqrt(0.15 * 0.85 / n)
    tol = 5.0 / np.sqrt(n)                                              # 5 sigma of a sample correlation of independent bits
    assert abs(_corr(lo, hi)) < tol                                     # the two halves of one draw
    for s in (1, 2, 8, 32):                                             # neighbouring elements of a row / rows of a tile
        assert abs(_corr(lo[:, s:], lo[:, :-s])) < tol and abs(_corr(hi[:, s:], hi[:, :-s])) < tol
    assert abs(_corr(lo[1:], lo[:-1])) < tol                            # the same element of consecutive DAGs
    other = rng.draw(rng.site_key((7 << 32) | 12345, 4, dags)[:, None], pairs[None, :])
    assert abs(_corr(lo, ((other & np.uint64(0xFFFF)) >= thr).astype(np.float64))) < tol      # another site, same DAGs
    # per-element and per-DAG keep rates scatter like binomials (no stuck elements, no lucky DAGs)
    assert abs(lo.mean(0).std() / np.sqrt(0.15 * 0.85 / lo.shape[0]) - 1) < 0.15
    assert abs(lo.mean(1).std() / np.sqrt(0.15 * 0.85 / lo.shape[1]) - 1) < 0.15


def test_pair_index_avalanche():
    keys = rng.site_key(99, 0, np.arange(2000, dtype=np.uint64))
    pairs = np.arange(512, dtype=np.uint64)
    h = rng.draw(keys[:, None], pairs[None, :])
    for b in range(9):
        x = h ^ rng.draw(keys[:, None], (pairs ^ np.uint64(1 << b))[None, :])
        flips = np.array([float(((x >> np.uint64(i)) & np.uint64(1)).mean()) for i in range(32)])
        assert flips.min() > 0.48 and flips.max() < 0.52, (b, flips.min(), flips.max())


def test_reparameterisation_noise_is_standard_normal():
    m = rng.DeviceMasks(2024, 0.15, eps_scale=1.0)
    e = m.eps(20000).astype(np.float64).ravel()
    n = e.size
    assert abs(e.mean()) < 5 / np.sqrt(n) and abs(e.var() - 1) < 5 * np.sqrt(2.0 / n)
    assert abs((e ** 3).mean()) < 5 * np.sqrt(15.0 / n) and abs((e ** 4).mean() - 3) < 5 * np.sqrt(96.0 / n)
    x = e.reshape(20000, -1)
    assert abs(_corr(x[:, 1:], x[:, :-1])) < 5 / np.sqrt(n) and abs(_corr(x[1:], x[:-1])) < 5 / np.sqrt(n)
