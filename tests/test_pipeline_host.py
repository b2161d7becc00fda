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


@pytest.mark.parametrize("use_polyco", [False, True, "tempo2"])
@pytest.mark.parametrize("turns,fractional", [(1.0, False), (3.0, False), (2.5, True)])
def test_turns_mode_divisions_match_oracle(oracle, use_polyco, turns, fractional):
    """dspsr -s / -turns N (TimeDivide.C:360-436,461-500): division boundaries from the pulse phase -- constant period and
    the vela.polyco predictor (phase / iphase with integer and fractional turns apart) -- against the oracle's restatement;
    the leading partial turn is skipped, every later sample lands in exactly one division."""
    from dspsr_amd import pipeline
    text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
    rate, t_start = 390625.0, 0.00108
    obs = oracle.Observation(tsamp_us=1e6 / rate)
    obs.start_seconds = t_start
    if use_polyco:
        if use_polyco == "tempo2":                                          # ChebyModelSet predictor, same interface
            import cheby_fixture as cf
            ctext = cf.cheby_text()
            pc, opc = pipeline.ChebyPredictor(ctext, 1400.0), oracle.ChebyPredictor(ctext, 1400.0)
        else:
            pc, opc = pipeline.Polyco(text), oracle.Polyco.parse(text)
        ocfg = oracle.FoldConfig(nbin=64, polyco=opc)
        assert pc.phase(55299, 7545.0 + 3.3) == oracle.predictor_phase(ocfg, obs, 3.3)
        phase = lambda t: pc.phase(55299, 7545.0 + t)
        iphase = lambda ph, guess: pc.iphase(ph, 55299, 7545.0 + guess) - 7545.0
        pguess = 1.0 / pc.frequency(55299, 7545.0 + t_start)
        # iphase inverts phase to well under a sample
        ph = pc.phase(55299, 7545.0 + 1.2345)
        assert abs(iphase(ph, 1.0) - 1.2345) < 1e-9
    else:
        period = 0.0893
        ocfg = oracle.FoldConfig(nbin=64, folding_period=period)
        phase = lambda t: (int(np.floor(t / period)), t / period - np.floor(t / period))
        iphase = lambda ph, guess: (ph[0] + ph[1]) * period
        pguess = period
    div = pipeline.TurnsDivider(phase, iphase, pguess, t_start, rate, turns, 0.0, fractional)
    for k in (0, 1, 2, 7):
        assert div.bounds(k) == oracle.subint_turns_sample_bounds(ocfg, obs, turns, k, fractional), k
    first = div.bounds(0)[0]
    if fractional:
        assert -int(pguess * rate) - 1 <= first <= 0                 # division 0 began before the data: nothing is skipped
    else:
        assert 0 < first <= int(pguess * rate) + 1                  # the leading partial turn is not folded
    pos, per_div = 0, {}
    for blk in (1000, 50000, 39070, 120000, 1, 200000):
        for idat, n, k, complete in div.pieces(pos, blk):
            lo, hi = div.bounds(k)
            assert lo <= pos + idat and pos + idat + n <= hi and complete == (pos + idat + n == hi)
            per_div[k] = per_div.get(k, 0) + n
        pos += blk
    for k, v in per_div.items():
        lo, hi = div.bounds(k)
        if k < max(per_div):
            assert v == hi - max(lo, 0)
            assert abs((hi - lo) - turns * pguess * rate) <= 2          # N turns long, to the sample


def test_normalise_profile_matches_oracle(oracle):
    from dspsr_amd import pipeline
    rng = np.random.default_rng(3)
    ps = oracle.PhaseSeries(3, 1, 4, 16)
    ps.data[...] = rng.standard_normal(ps.data.shape).astype(np.float32)
    ps.hits[:] = rng.integers(0, 5, 16)
    ps.hits[0] = 0
    got = pipeline.normalise_profile(ps.data, ps.hits, 4096.0 * 4194304.0)
    assert np.array_equal(got, oracle.archive_profile(ps, 4096.0 * 4194304.0))


def test_phase_series_file_round_trip(tmp_path):
    """Archive hand-off: the sub-integration file carries hits + un-normalised sums + the observation keys."""
    from dspsr_amd import pipeline
    rng = np.random.default_rng(3)
    cfg = pipeline.Config(nchan=8, dispersion_measure=12.5, nbin=32, folding_period=0.0893, ndim=4)
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-16.0)
    sub = {"hits": rng.integers(0, 100, 32).astype(np.uint32), "integration_length": 1.25, "ndat_total": 12345,
           "profile": rng.standard_normal((8, 1, 32, 4)).astype(np.float32)}
    path = str(tmp_path / "sub0.ps")
    pipeline.write_phase_series(path, sub, info, cfg, scale=4096.0 * 32, division=3, start_seconds=0.5,
                                folding_period=0.0893)
    hdr, hits, prof = pipeline.read_phase_series(path)
    assert hdr["HDR_MAGIC"] == pipeline.PHASE_SERIES_MAGIC and int(hdr["HDR_SIZE"]) == 4096
    assert (float(hdr["FREQ"]), float(hdr["BW"]), float(hdr["DM"])) == (1382.0, -16.0, 12.5)
    assert (int(hdr["DIVISION"]), int(hdr["NDAT_TOTAL"]), float(hdr["INTEGRATION_LENGTH"])) == (3, 12345, 1.25)
    assert hdr["STATE"] == "Coherence" and float(hdr["SCALE"]) == 4096.0 * 32
    assert np.array_equal(hits, sub["hits"]) and np.array_equal(prof, sub["profile"])
    with open(path, "r+b") as f:
        f.truncate(4096 + 4 * 32 + 10)
    with pytest.raises(pipeline.DspsrAmdError, match="truncated"):
        pipeline.read_phase_series(path)
    bad = tmp_path / "bad.ps"
    bad.write_bytes(b"HDR_MAGIC something\n".ljust(4096, b"\0"))
    with pytest.raises(pipeline.DspsrAmdError, match="is not a"):
        pipeline.read_phase_series(str(bad))


def test_combine_phase_series_matches_oracle_fold_of_time_slices(oracle):
    """PhaseSeries::combine semantics: two replicas folding consecutive time slices of the same division, merged,
    equal one fold of the whole span (hits exactly; sums to float rounding)."""
    from dspsr_amd import pipeline
    o = oracle
    rng = np.random.default_rng(4)
    nchan, nbin, ndat = 4, 16, 4000
    det = rng.standard_normal((nchan, 1, ndat, 4)).astype(np.float32)
    obs = o.Observation(centre_frequency=1382.0, bandwidth=-4.0, tsamp_us=1.0)
    cfg = o.FoldConfig(nbin=nbin, folding_period=0.000731)
    parts = []
    for a, b in ((0, 1500), (1500, 4000)):
        ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float32))
        o.fold(det, obs, cfg, ps, idat_start=a, ndat_fold=b - a)
        parts.append({"hits": ps.hits.copy(), "integration_length": ps.integration_length, "ndat_total": ps.ndat_total,
                      "profile": ps.data.copy()})
    whole = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    o.fold(det, obs, cfg, whole, idat_start=0, ndat_fold=1500)
    o.fold(det, obs, cfg, whole, idat_start=1500, ndat_fold=2500)
    merged = pipeline.combine_phase_series(None, parts[0])
    assert merged["integration_length"] == parts[0]["integration_length"]         # empty + b = copy of b
    merged = pipeline.combine_phase_series(merged, parts[1])
    assert np.array_equal(merged["hits"], whole.hits) and merged["ndat_total"] == ndat
    assert abs(merged["integration_length"] - whole.integration_length) < 1e-12
    assert np.abs(merged["profile"] - whole.data).max() <= 1e-5 * np.abs(whole.data).max()
    with pytest.raises(pipeline.DspsrAmdError, match="mixable"):
        pipeline.combine_phase_series(merged, {"hits": np.zeros(8, np.uint32), "integration_length": 1.0, "ndat_total": 1,
                                               "profile": np.zeros((nchan, 1, 8, 4), np.float32)})


def test_sigproc_header_bytes():
    """filterbank_header.c / send_stuff.c: length-prefixed keywords, int32 and float64 values, HEADER_START..HEADER_END."""
    import io
    import struct
    from dspsr_amd import pipeline
    f = io.BytesIO()
    pipeline.write_sigproc_header(f, source_name="J0835-4510", fch1=1581.951171875, foff=-0.09765625, nchans=4096, nbits=8,
                                  tstart_mjd=55299.087326388886, tsamp=0.00016384)
    b = f.getvalue()
    assert b.startswith(struct.pack("<i", 12) + b"HEADER_START") and b.endswith(struct.pack("<i", 10) + b"HEADER_END")

    def value(key, fmt):
        k = struct.pack("<i", len(key)) + key.encode()
        i = b.index(k) + len(k)
        return struct.unpack_from(fmt, b, i)[0]
    assert value("nchans", "<i") == 4096 and value("nbits", "<i") == 8 and value("nifs", "<i") == 1 and value("data_type", "<i") == 1
    assert value("fch1", "<d") == 1581.951171875 and value("foff", "<d") == -0.09765625 and value("tsamp", "<d") == 0.00016384
    assert value("tstart", "<d") == 55299.087326388886
    i = b.index(b"source_name") + len("source_name")
    assert b[i:i + 4] == struct.pack("<i", 10) and b[i + 4:i + 14] == b"J0835-4510"


def test_sigproc_header_matches_the_reference_writer(tmp_path):
    """tests/golden/sigproc_header.bin was written by the reference's own filterbank_header.c / send_stuff.c (compiled
    unmodified into oracle/_ref/libsigproc_ref.so, generator tests/golden/make_sigproc_header.py): our writer must produce
    the same bytes; when the reference build is present the comparison is repeated live with other values."""
    import io
    import importlib.util
    from dspsr_amd import pipeline
    gold = os.path.join(ROOT, "tests", "golden")
    values = json.load(open(os.path.join(gold, "sigproc_header.json")))
    f = io.BytesIO()
    pipeline.write_sigproc_header(f, **values)
    assert f.getvalue() == open(os.path.join(gold, "sigproc_header.bin"), "rb").read()
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsigproc_ref.so")):
        spec = importlib.util.spec_from_file_location("make_sigproc_header", os.path.join(gold, "make_sigproc_header.py"))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        v2 = dict(values, source_name="B1937+21", fch1=1500.0, foff=-0.5, nchans=512, nbits=2, tsamp=6.4e-5, tstart_mjd=60000.25)
        ref = gen.reference_header_bytes(v2, str(tmp_path / "h.bin"))
        f2 = io.BytesIO()
        pipeline.write_sigproc_header(f2, **v2)
        assert f2.getvalue() == ref


@pytest.mark.parametrize("use_polyco", [False, True])
@pytest.mark.parametrize("turns,ref", [(0.25, 0.0), (0.1, 0.35), (0.5, 0.9)])
def test_phase_resolved_divisions(oracle, use_polyco, turns, ref):
    """Divisions shorter than one turn (TimeDivide.C:374-425): they start at the first boundary reference_phase + N*D after
    the phase of the first sample, and follow each other every D turns; against the oracle's restatement, and every sample
    after the start lands in exactly one division."""
    from dspsr_amd import pipeline
    text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
    rate, t_start = 390625.0, 0.00108
    obs = oracle.Observation(tsamp_us=1e6 / rate)
    obs.start_seconds = t_start
    if use_polyco:
        pc, opc = pipeline.Polyco(text), oracle.Polyco.parse(text)
        ocfg = oracle.FoldConfig(nbin=64, polyco=opc, reference_phase=ref)
        phase = lambda t: pc.phase(55299, 7545.0 + t)
        iphase = lambda ph, guess: pc.iphase(ph, 55299, 7545.0 + guess) - 7545.0
        pguess = 1.0 / pc.frequency(55299, 7545.0 + t_start)
    else:
        period = 0.0893
        ocfg = oracle.FoldConfig(nbin=64, folding_period=period, reference_phase=ref)
        phase = lambda t: (int(np.floor(t / period)), t / period - np.floor(t / period))
        iphase = lambda ph, guess: (ph[0] + ph[1]) * period
        pguess = period
    div = pipeline.TurnsDivider(phase, iphase, pguess, t_start, rate, turns, ref)
    (spi, spf), st = oracle.subint_turns_start(ocfg, obs, turns)
    assert div.start_phase[0] == spi and abs(div.start_phase[1] - spf) < 1e-12 and abs(div.start_time - st) < 1e-9
    # the start is a division boundary (fractional phase = ref + N*D) less than one division after the first sample
    n = (spf - ref) / turns
    assert abs(n - round(n)) < 1e-9 or abs((spf + 1 - ref) / turns - round((spf + 1 - ref) / turns)) < 1e-9
    assert 0 <= div.bounds(0)[0] <= int(turns * pguess * rate) + 1
    for k in (0, 1, 5, 40):
        assert div.bounds(k) == oracle.subint_turns_sample_bounds(ocfg, obs, turns, k), k
        lo, hi = div.bounds(k)
        assert abs((hi - lo) - turns * pguess * rate) <= 1.5
    pos, seen = 0, 0
    for blk in (700, 30000, 12345, 1, 60000):
        for idat, cnt, k, complete in div.pieces(pos, blk):
            lo, hi = div.bounds(k)
            assert lo <= pos + idat and pos + idat + cnt <= hi and complete == (pos + idat + cnt == hi)
            seen += cnt
        pos += blk
    assert seen == pos - div.bounds(0)[0]


def test_cheby_predictor_closed_form_and_oracle(oracle):
    """TEMPO2 predictor (ChebyModelSet): the product's evaluator and the oracle's independent one against the closed form of
    the phase law the synthetic predictor text was fitted to (tests/cheby_fixture.py; tempo2 itself is absent: parity
    unpinned) -- integer and fractional turns apart at 1.2e10 turns, spin frequency, two segments, iphase round trip."""
    import cheby_fixture as cf
    from dspsr_amd import pipeline
    text = cf.cheby_text()
    for f in (None, 1400.0, 1200.5):
        a, b = pipeline.ChebyPredictor(text, f), oracle.ChebyPredictor(text, f)
        fobs = f if f is not None else 1382.0                                  # default: the middle of FREQ_RANGE
        assert a.observing_frequency == b.observing_frequency == fobs
        for sec in (0.05 * 86400 + 0.5, 7545.0, 7545.0 + 3.3, 0.15 * 86400 - 1.0, 0.15 * 86400 + 1.0, 0.2 * 86400 + 17.0):
            ph, fr = cf.closed_form(sec, fobs)
            wi, wf = int(ph // 1), float(ph - ph // 1)
            for pr in (a, b):
                gi, gf = pr.phase(55299, sec)
                assert abs((gi - wi) + (gf - wf)) <= 1e-9, (sec, gi, gf, wi, wf)
                assert abs(pr.frequency(55299, sec) - fr) <= 1e-11 * fr
                assert pr.phase_frac(55299, sec) == gf
            ai, af = a.phase(55299, sec)
            bi, bf = b.phase(55299, sec)
            assert abs((ai - bi) + (af - bf)) <= 1e-10
        t = a.iphase((a.phase(55299, 7545.0)[0] + 3, 0.25), 55299, 7545.0)
        gi, gf = a.phase(55299, t)
        assert gi == a.phase(55299, 7545.0)[0] + 3 and abs(gf - 0.25) <= 1e-9
    with pytest.raises(pipeline.DspsrAmdError):
        pipeline.ChebyPredictor(text).phase(55299, 0.3 * 86400)                    # outside every TIME_RANGE
    with pytest.raises(pipeline.DspsrAmdError):
        pipeline.ChebyPredictor("ChebyModelSet 0 segments\n")
