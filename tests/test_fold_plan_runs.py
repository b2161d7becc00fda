"""The fold plan found run by run (csrc/host_prep.cpp fold_plan_run, used by dspsr_amd_fold_set_bins) against the sample-by-sample
recurrence of Fold.C:744-787 (dspsr_amd_fold_binplan == oracle.fold_binplan, tests/test_capi.py): the same runs and the same hits for
random start phases, steps and bin counts, and for the cases the short cut must decline (ties, steps below the spacing of the doubles,
steps of a turn and more, negative steps, zero).  Host code only: runs without a GPU."""
import numpy as np
import pytest

import dspsr_amd


def _runs_of(plan):
    if plan.size == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint64)
    cut = np.flatnonzero(np.diff(plan.astype(np.int64)) != 0) + 1
    off = np.concatenate([[0], cut]).astype(np.uint64)
    return off, plan[off.astype(np.int64)], np.diff(np.concatenate([off, [plan.size]]).astype(np.int64)).astype(np.uint64)


def _check(phi, pps, nbin, ndat):
    plan, hits = dspsr_amd.fold_binplan(phi, pps, nbin, ndat)
    off, rb, rh, hits2, n = dspsr_amd.fold_binplan_runs(phi, pps, nbin, ndat)
    woff, wb, wh = _runs_of(plan)
    assert n == woff.size, (phi, pps, nbin, ndat, n, woff.size)
    assert np.array_equal(off, woff) and np.array_equal(rb, wb) and np.array_equal(rh, wh), (phi, pps, nbin, ndat)
    assert np.array_equal(hits, hits2)


def test_random_plans_equal_the_sample_loop():
    rng = np.random.default_rng(2026)
    for i in range(4000):
        nbin = int(rng.choice([1, 2, 3, 16, 64, 100, 512, 1000, 1024, 4096, 65536, int(rng.integers(1, 100000))]))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            pps = float(10.0 ** rng.uniform(-9, -1))                       # the usual: a small fraction of a turn per sample
        elif kind == 1:
            pps = float(rng.uniform(0, 1)) / nbin / float(rng.integers(1, 2000))   # runs of about that many samples
        elif kind == 2:
            pps = float(np.ldexp(float(rng.integers(1, 1 << 20)), -int(rng.integers(20, 75))))     # few significant bits: exact sums, ties
        elif kind == 3:
            pps = float(rng.uniform(0.3, 3.0))                              # a good part of a turn and more per sample
        elif kind == 4:
            pps = float(10.0 ** rng.uniform(-22, -12))                      # below the spacing of the doubles near 1: the phase sticks
        else:
            pps = 1.0 / (float(rng.integers(2, 5000)) + float(rng.uniform(0, 1)))
        phi = float(rng.choice([0.0, rng.uniform(0, 1), 1.0 - 2.0 ** -53, rng.uniform(-3, 3), 0.5, 2.0 ** -int(rng.integers(1, 60))]))
        ndat = int(rng.choice([1, 2, 17, 1000, 20000, 200000]))
        _check(phi, pps, nbin, ndat)


@pytest.mark.parametrize("phi,pps,nbin,ndat", [
    (0.731, 1.0 / 345.67, 1024, 100000), (0.0, 0.0, 64, 1000), (0.25, -0.001, 128, 5000), (0.999999, 2.0 ** -53, 1024, 4000),
    (0.5, 2.0 ** -53 * 1.5, 512, 5000),                  # a tie in [0.5, 1): every addition rounds to even -- the sample loop decides
    (0.75, 2.0 ** -54, 512, 3000), (0.1, 1.0, 16, 100), (0.1, 7.25, 16, 100), (1e-300, 1e-3, 256, 5000), (0.0, 2.0 ** -30, 1 << 16, 300000),
    (0.123456789, 1.0 / 8720.123, 512, 3000000),         # the after8c workload's shape: runs of 8720 samples
    (0.9, 1.0 / 3.0, 3, 1000), (0.3, 0.1, 10, 1000),
])
def test_chosen_plans_equal_the_sample_loop(phi, pps, nbin, ndat):
    _check(phi, pps, nbin, ndat)
