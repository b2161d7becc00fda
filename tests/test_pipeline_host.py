"""CPU tests of the product's host-side pipeline logic against the oracle restatements
(a9 choose_nbin, a10 predictor phase, a12 sub-integration division, a14 archive normalisation)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("period,rate,req", [(0.0893, 390625.0, 0), (0.0015, 390625.0, 0), (0.0893, 390625.0, 512),
                                             (1e-5, 390625.0, 0), (0.005, 1e6, 0), (0.7, 31250.0, 2048)])
def test_choose_nbin(oracle, period, rate, req):
    from dspsr_amd import pipeline
    assert pipeline.choose_nbin(period, rate, req) == oracle.choose_nbin(period, rate, req)
    assert pipeline.choose_nbin(period, rate, req, force_sensible_nbin=True) == \
        oracle.choose_nbin(period, rate, req, force_sensible_nbin=True)


def test_choose_nbin_invalid_period():
    from dspsr_amd import pipeline, DspsrAmdError
    with pytest.raises(DspsrAmdError, match="invalid folding period"):
        pipeline.choose_nbin(0.0, 1e6)


def test_polyco_matches_oracle(oracle):
    from dspsr_amd import pipeline
    text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
    a, b = pipeline.Polyco(text), oracle.Polyco.parse(text)
    for dt in (0.0, 0.123456, 17.5, 1800.0, 3599.9):
        assert a.phase_frac(55299, 7545.0 + dt) == b.phase_frac(55299, 7545.0 + dt)
        assert a.frequency(55299, 7545.0 + dt) == b.frequency(55299, 7545.0 + dt)


def test_subint_pieces_cover_block_and_match_oracle_boundaries(oracle):
    from dspsr_amd import pipeline
    obs = oracle.Observation(tsamp_us=2.56)
    for L in (0.0101, 0.0033, 0.25):      # (k*L*rate hitting exactly .5 is a rounding quirk of the reference)
        pos = 0
        per_div = {}
        for blk in (1000, 5000, 3907, 12000, 1, 200000):
            pieces = pipeline.subint_pieces(pos, blk, L, obs.rate)
            assert sum(p[1] for p in pieces) == blk and pieces[0][0] == 0
            for idat, n, div, complete in pieces:
                lo, hi = oracle.subint_sample_bounds(obs, L, div)
                # the reference's two roundings can leave a one-sample gap between upper(k) and lower(k+1);
                # the product gives that sample to division k+1
                assert lo - 1 <= pos + idat and pos + idat + n <= hi + 1
                assert complete == (pos + idat + n == hi)
                per_div[div] = per_div.get(div, 0) + n
            pos += blk
        for d, v in per_div.items():
            lo, hi = oracle.subint_sample_bounds(obs, L, d)
            if d < max(per_div):
                assert abs(v - (hi - lo)) <= 1


def test_normalise_profile_matches_oracle(oracle):
    from dspsr_amd import pipeline
    rng = np.random.default_rng(3)
    ps = oracle.PhaseSeries(3, 1, 4, 16)
    ps.data[...] = rng.standard_normal(ps.data.shape).astype(np.float32)
    ps.hits[:] = rng.integers(0, 5, 16)
    ps.hits[0] = 0
    got = pipeline.normalise_profile(ps.data, ps.hits, 4096.0 * 4194304.0)
    assert np.array_equal(got, oracle.archive_profile(ps, 4096.0 * 4194304.0))
