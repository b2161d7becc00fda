"""GPU parity of the NON-CONVOLVING filterbank: dsp::Filterbank with freq_res = 1 (`dspsr -F N` -- Filterbank::Config::After /
Never; Signal/General/Filterbank.C:561-662 with the `freq_res == 1` branch :614-623; GPU twin FilterbankCUDA.cu:92-116,258-304 with
plan_bwd == NULL).  The HIP kernel (csrc/fb_plain.hip, through the C-ABI) against the float64 oracle restatement
(oracle.filterbank, M == 1) on the same seeded input, every input form the convolving filterbank takes.

Tolerance: that of tests/test_gpu_parity.py (raw filterbank output vs the float64 oracle: rms(err)/rms(out) <= 2e-6 * sqrt(log2 2N),
max |err| <= 8 times that); detected / folded output <= 1e-5 of the largest value per the north star."""
import math

import numpy as np
import pytest

from test_gpu_parity import _fb_case, _raw

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import dspsr_amd
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    yield dspsr_amd, ctx
    ctx.close()


# (C, npart): tiles of 2^13 points hold 2^12 / C parts of two polarisations (2^14 points from 2048 channels on); the part counts
# leave ragged last tiles and cover one tile, several tiles and more tiles than workgroups
@pytest.mark.parametrize("C,npart", [(2, 700), (4, 1030), (8, 513), (16, 300), (32, 131), (64, 200), (128, 70), (256, 37),
                                     (512, 9), (1024, 11), (2048, 6), (4096, 3), (8192, 2), (128, 40000)])
def test_plain_filterbank_real_dualpol_raw(oracle, gpu, C, npart):
    _fb_case(oracle, gpu, C, 1, (0, 0), npart)


@pytest.mark.parametrize("C,npart", [(16, 77), (128, 100), (1024, 7), (4096, 2)])
def test_plain_filterbank_float_rows_equal_raw(oracle, gpu, C, npart):
    """Filterbank::Engine::perform is handed unpacked float32 rows (FilterbankEngine.h:28-32)."""
    a, _ = _fb_case(oracle, gpu, C, 1, (0, 0), npart, use_raw=True)
    b, _ = _fb_case(oracle, gpu, C, 1, (0, 0), npart, use_raw=False)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()


def test_plain_filterbank_caspsr_layout(oracle, gpu):
    _fb_case(oracle, gpu, 128, 1, (0, 0), 50, layout="caspsr")
    _fb_case(oracle, gpu, 2048, 1, (0, 0), 5, layout="caspsr")


@pytest.mark.parametrize("C,npart", [(4, 300), (64, 130), (512, 17), (8192, 3)])
def test_plain_filterbank_single_pol(oracle, gpu, C, npart):
    _fb_case(oracle, gpu, C, 1, (0, 0), npart, npol=1)
    _fb_case(oracle, gpu, C, 1, (0, 0), npart, npol=1, use_raw=False)


@pytest.mark.parametrize("input_nchan,npol", [(1, 2), (4, 2), (2, 1), (8, 2)])
@pytest.mark.parametrize("C,npart", [(8, 301), (128, 67), (2048, 5)])
def test_plain_filterbank_complex_input(oracle, gpu, input_nchan, npol, C, npart):
    """Signal::Analytic input: fcc1d of nchan_subband complex samples per part (Filterbank.C:593); several input channels: output
    channel input_ichan * nchan_subband + k (Filterbank.C:619)."""
    _fb_case(oracle, gpu, C, 1, (0, 0), npart, npol=npol, real=False, input_nchan=input_nchan)
    _fb_case(oracle, gpu, C, 1, (0, 0), npart, npol=npol, real=False, input_nchan=input_nchan, use_raw=False)


def test_plain_filterbank_real_multichannel_input(oracle, gpu):
    _fb_case(oracle, gpu, 64, 1, (0, 0), 90, input_nchan=3)
    _fb_case(oracle, gpu, 64, 1, (0, 0), 90, input_nchan=3, use_raw=False)


def _engine_case(oracle, gpu, C, npart, npol=2, real=True, kernel=None, seed=5):
    dspsr_amd, ctx = gpu
    o = oracle
    obs = o.Observation(nchan=1, npol=npol, ndim=1 if real else 2)
    plan = o.filterbank_plan(obs, C, None, 1)
    raw = _raw(npart * plan.nsamp_step, npol, obs.ndim, 1, seed)
    unpacked = o.unpack_8bit(raw, obs)
    ref = o.filterbank(unpacked, plan, kernel, npart=npart, dtype=np.float64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, 1, 0, 0, 1, npol, real, kernel)
    assert (eng.nsamp_fft, eng.nsamp_overlap, eng.nsamp_step, eng.nkeep) == (plan.nsamp_fft, 0, plan.nsamp_step, 1)
    assert eng.npass() == 1 and eng.fold_is_fused() == 0 and not eng.search_is_fused()
    return eng, torch.from_numpy(raw).cuda(), ref, float(o.S8)


def test_plain_filterbank_without_response(oracle, gpu):
    """`dspsr -F N` sets no response on the filterbank (LoadToFold1.C:318-323: only for Config::During)."""
    dspsr_amd, ctx = gpu
    eng, d_raw, ref, scale = _engine_case(oracle, gpu, 256, 45)
    out = torch.zeros((256, 2, 2 * 45), dtype=torch.float32, device="cuda")
    eng.perform_raw(d_raw, dspsr_amd.RAW_GENERIC, scale, out, 45)
    eng.finish()
    got = out.cpu().numpy().view(np.complex64).astype(np.complex128)
    eng.close()
    rms = math.sqrt(np.mean(np.abs(ref) ** 2))
    tol = 2e-6 * math.sqrt(math.log2(512))
    assert math.sqrt(np.mean(np.abs(got - ref) ** 2)) / rms <= tol
    assert np.abs(got - ref).max() <= 8 * tol * rms


def test_plain_filterbank_strided_output_rows(oracle, gpu):
    """out_step > 2 * nkeep and rows with head room (TimeSeries::resize reserves space in front of the data, TimeSeries.C:146-179):
    the samples land out_step floats apart and nothing else is written."""
    dspsr_amd, ctx = gpu
    C, npart = 64, 75
    eng, d_raw, ref, scale = _engine_case(oracle, gpu, C, npart)
    big = torch.full((C, 2, 4 * npart + 10), 7.0, dtype=torch.float32, device="cuda")
    view = big[:, :, 6:]
    eng.perform_raw(d_raw, dspsr_amd.RAW_GENERIC, scale, view, npart, out_step=4)
    eng.finish()
    h = big.cpu().numpy()
    eng.close()
    got = (h[:, :, 6:6 + 4 * npart].reshape(C, 2, npart, 4)[..., 0] + 1j * h[:, :, 6:6 + 4 * npart].reshape(C, 2, npart, 4)[..., 1])
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert (h[:, :, :6] == 7.0).all() and (h[:, :, 6:6 + 4 * npart].reshape(C, 2, npart, 4)[..., 2:] == 7.0).all()
    assert (h[:, :, 6 + 4 * npart:] == 7.0).all()


@pytest.mark.parametrize("state", ["Coherence", "Stokes"])
@pytest.mark.parametrize("ndim", [1, 2, 4])
def test_plain_filterbank_detected_output(oracle, gpu, state, ndim):
    """perform_detect on the non-convolving filterbank == Filterbank, then Detection::polarimetry (Detection.C:322-474 layouts)."""
    dspsr_amd, ctx = gpu
    C, npart = 128, 333
    eng, d_raw, ref, scale = _engine_case(oracle, gpu, C, npart)
    want = oracle.detect_layout(oracle.detect_products(ref, state), ndim)
    det = torch.zeros((C, 4 // ndim, npart * ndim), dtype=torch.float32, device="cuda")
    st = dspsr_amd.COHERENCE if state == "Coherence" else dspsr_amd.STOKES
    eng.perform_detect(det, npart, st, ndim, raw=d_raw, scale=scale)
    eng.finish()
    got = det.cpu().numpy().reshape(want.shape if ndim > 1 else (C, 4, npart))
    eng.close()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


def test_plain_filterbank_detect_and_fold(oracle, gpu):
    """`dspsr -F N` without dedispersion (Config::Never): Filterbank -> Detection -> Fold through perform_fold == the same three
    operations one by one, bit for bit, and the float64 oracle to 1e-5."""
    dspsr_amd, ctx = gpu
    C, npart, nbin = 256, 3000, 64
    eng, d_raw, ref, scale = _engine_case(oracle, gpu, C, npart)
    phi, pps = 0.23, 1.0 / 91.7
    plan = oracle.fold_binplan(phi, pps, nbin, npart)
    prod = oracle.detect_products(ref, "Coherence")                  # [chan][4][ndat] float64
    want = np.zeros((C, nbin, 4))
    for k in range(4):
        for c in range(C):
            want[c, :, k] = np.bincount(plan, weights=prod[c, k], minlength=nbin)
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(C, 1, 4, nbin)
    hits = np.zeros(nbin, np.uint32)
    fold.set_nbin(nbin)
    fold.set_ndat(npart, 0)
    fold.set_bins(phi, pps, npart, 0, hits)
    eng.perform_fold(fold, npart, raw=d_raw, scale=scale)
    got = fold.synch().reshape(C, nbin, 4)
    assert np.array_equal(hits, np.bincount(plan, minlength=nbin).astype(np.uint32))
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    # the same through Detection + Fold on a detected block
    det = torch.zeros((C, 1, npart * 4), dtype=torch.float32, device="cuda")
    eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=d_raw, scale=scale)
    fold2 = dspsr_amd.FoldEngine(ctx)
    fold2.set_shape(C, 1, 4, nbin)
    fold2.set_nbin(nbin)
    fold2.set_ndat(npart, 0)
    fold2.set_bins(phi, pps, npart, 0, np.zeros(nbin, np.uint32))
    fold2.fold(det)
    assert np.array_equal(fold2.synch().reshape(C, nbin, 4), got)
    fold.close(); fold2.close(); eng.close()


def test_plain_filterbank_search_mode(oracle, gpu):
    """digifil's Filterbank branch without a response and without -x (LoadToFil.C:185-222: `filterbank->set_nchan`, freq_res 1):
    Filterbank -> Detection::square_law -> TScrunch, as a stream over two calls, bit for bit against the float32 loop on the
    filterbank's own output."""
    dspsr_amd, ctx = gpu
    C, npart, sf = 128, 1001, 16
    eng, d_raw, ref, scale = _engine_case(oracle, gpu, C, npart)
    cplx = torch.zeros((C, 2, 2 * npart), dtype=torch.float32, device="cuda")
    eng.perform_raw(d_raw, dspsr_amd.RAW_GENERIC, scale, cplx, npart)
    fb32 = cplx.cpu().numpy().view(np.complex64)
    want = oracle.tscrunch_fpt(oracle.square_law(fb32, "Intensity"), sf)
    out = torch.zeros((C, 1, npart // sf + 1), dtype=torch.float32, device="cuda")
    carry = torch.zeros((C, 1), dtype=torch.float32, device="cuda")
    n1 = 600
    nout1, cc = eng.perform_search(out, carry, 0, n1, sf, raw=d_raw, scale=scale)
    rest = d_raw[n1 * eng.nsamp_step * 2:]
    nout2, cc = eng.perform_search(out[:, :, nout1:], carry, cc, npart - n1, sf, raw=rest, scale=scale)
    eng.finish()
    assert nout1 + nout2 == npart // sf and cc == npart % sf
    assert np.array_equal(out.cpu().numpy()[:, :, :nout1 + nout2], want)
    eng.close()


def test_plain_filterbank_refusals(gpu):
    dspsr_amd, ctx = gpu
    for bad in (dict(nchan_subband=1), dict(nchan_subband=96), dict(nchan_subband=16384), dict(nfilt_pos=1)):
        kw = dict(nchan_subband=64, nfilt_pos=0)
        kw.update(bad)
        with pytest.raises(dspsr_amd.DspsrAmdError):
            dspsr_amd.FilterbankEngine(ctx).setup(kw["nchan_subband"], 1, kw["nfilt_pos"], 0)
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.FilterbankEngine(ctx).setup(64, 0, 0, 0)                       # Response.ndat = 0 (Filterbank.C:101-103)


def _after_oracle(o, raw, obs, nchan, nbin, period, when, kernel=None):
    """Oracle chain of `dspsr -F N` (Config::After): Filterbank with freq_res = 1, Dedispersion matched to ITS OUTPUT, Convolution,
    Detection, Fold -- in float64."""
    plan1 = o.filterbank_plan(obs, nchan, None, 1)
    fb1 = o.filterbank(o.unpack_8bit(raw, obs), plan1, None, dtype=np.float64)
    obs1 = o.filterbank_output_observation(obs, plan1)
    fobs = o.Observation(**obs1.__dict__)
    resp = None
    if when == "after":
        resp = o.Dedispersion().match(obs1)
        if kernel is not None:
            # the product's host-built chirp (std::polar(float)) == the oracle's (float64 cos / sin, rounded) to one ulp
            assert np.abs(resp.buffer - kernel).max() <= 1.2e-7
        fb1 = o.convolution(np.ascontiguousarray(fb1).view(np.float64), resp.ndat, resp.impulse_pos, resp.impulse_neg, resp.buffer, False,
                            dtype=np.float64)
        fobs.start_seconds += resp.impulse_pos / fobs.rate         # Convolution.C:300
    det = o.detect_layout(o.detect_products(fb1, "Coherence"), 4)
    return det, fobs, resp


@pytest.mark.parametrize("when", ["after", "never"])
@pytest.mark.parametrize("machine", ["DADA", "CASPSR"])
def test_pipeline_filterbank_then_convolution(oracle, gpu, when, machine):
    """`dspsr -F 16` on a dispersed pulsed signal: non-convolving filterbank, then dsp::Convolution per channel with the response matched
    to the filterbank's output (DC-centred dual-sideband channels), Detection, Fold -- several blocks, against the float64 oracle."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    o = oracle
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=3,
                          max_parts=2, convolve_when=when)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine=machine)
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    nblocks = 3
    step = cfg.parts_per_block * lt.nsamp_step
    ndat = nblocks * step + lt.nsamp_overlap
    ndat += (-ndat) % 4                                           # (CASPSR blocks hold whole groups of four samples)
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period, layout="caspsr" if machine == "CASPSR" else "generic")
    d_raw = torch.from_numpy(raw).cuda()
    for b in range(nblocks):
        lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.finish_subint()
    lt.synchronize()
    got = lt.subints[0]["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4)
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm, machine=machine)
    det, fobs, resp = _after_oracle(o, raw, obs, nchan, nbin, period, when, lt.response.kernel if when == "after" else None)
    ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    per_block = cfg.parts_per_block * lt.nkeep
    for b in range(nblocks):
        o.fold(det, fobs, fcfg, ps, idat_start=b * per_block, ndat_fold=per_block)
    assert np.array_equal(lt.subints[0]["hits"], ps.hits)
    assert np.abs(got - ps.data).max() <= 1e-5 * np.abs(ps.data).max()
    assert abs(lt.out_rate - fobs.rate) <= 1e-9 * fobs.rate and abs(lt.out_start - fobs.start_seconds) <= 1e-12
    lt.close()


@pytest.mark.parametrize("nchan,M,nfilt,npart", [(16, 2048, (200, 150), 5), (128, 1024, (100, 90), 3), (12, 512, (40, 30), 7),
                                                 (8, 65536, (3000, 2000), 2), (4, 256, (20, 13), 40), (3, 8192, (700, 650), 5),
                                                 (5, 4096, (0, 0), 3), (2, 64, (5, 7), 300), (1, 128, (9, 0), 130),
                                                 (3, 16384, (1000, 900), 4), (2, 32768, (0, 0), 2), (2, 131072, (5000, 4000), 4),
                                                 (2, 262144, (100, 200), 2), (1, 524288, (1000, 1000), 2), (2, 1048576, (30000, 20000), 2),
                                                 (3, 2097152, (5, 100000), 2), (1, 4194304, (1000, 1000), 1)])
def test_convolution_of_many_channels_as_one_launch_group(oracle, gpu, nchan, M, nfilt, npart):
    """dsp::Convolution behind a filterbank (`dspsr -F N`: nchan_subband = 1 on many complex channels, float32 rows): the channels of a
    group run as ONE launch group -- forward passes per channel, inverse passes of a group-wide filterbank (filterbank.hip
    fb_run_batched).  Against the float64 oracle, and bit for bit against the loop over the channels (force_four_pass = 2 turns the
    grouping off); complex rows and detected output; parts in several launch groups."""
    dspsr_amd, ctx = gpu
    o = oracle
    rng = np.random.default_rng(31)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, nchan * M)).astype(np.complex64)
    step = M - sum(nfilt)
    x = rng.standard_normal((nchan, 2, 2 * (npart * step + sum(nfilt)))).astype(np.float32)
    ref = o.convolution(x, M, nfilt[0], nfilt[1], kernel, False, npart=npart, dtype=np.float64)
    d_x = torch.from_numpy(x).cuda()
    outs, dets = [], []
    # force_four_pass 1: four tile passes, channels grouped; 2: four tile passes, loop over the channels; 0: the library's choice
    # (one tile pass for n_fft <= 8192, csrc/fb_conv1.hip; three for 2^14 ... 2^21, csrc/fb_conv3.hip; grouped four-pass above)
    # (n_fft = 2^17 with max_parts = 1024: the three-pass scratch then holds one channel per launch group -- the loop over channel groups;
    #  n_fft = 2^21 with max_parts = 30: two channels per group, three channels -- a ragged last group)
    for ffp in (1, 2, 0):
        eng = dspsr_amd.FilterbankEngine(ctx).setup(1, M, nfilt[0], nfilt[1], nchan, 2, False, kernel, max_parts=1024 if M == 131072 else 30 if M == 2097152 else 3,
                                                    force_four_pass=ffp)
        assert eng.npass(False) == ((1 if M <= 8192 else 3 if M <= 2097152 else 4) if ffp == 0 else 4)
        out = torch.zeros((nchan, 2, 2 * npart * step), dtype=torch.float32, device="cuda")
        eng.perform(d_x, out, npart, 2 * step, 2 * step)
        det = torch.zeros((nchan, 1, 4 * npart * step), dtype=torch.float32, device="cuda")
        eng.perform_detect(det, npart, dspsr_amd.STOKES, 4, inp=d_x, in_step=2 * step)
        eng.finish()
        outs.append(out.cpu().numpy())
        dets.append(det.cpu().numpy())
        eng.close()
    rms = math.sqrt(np.mean(np.abs(ref) ** 2))
    tol = 2e-6 * math.sqrt(math.log2(2 * M))
    want = oracle.detect_layout(oracle.detect_products(ref, "Stokes"), 4).reshape(nchan, 1, -1)
    for o_, d_ in zip(outs, dets):
        got = o_.view(np.complex64).astype(np.complex128)
        assert math.sqrt(np.mean(np.abs(got - ref) ** 2)) / rms <= tol
        assert np.abs(got - ref).max() <= 8 * tol * rms
        assert np.abs(d_ - want).max() <= 1e-5 * np.abs(want).max()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(dets[0], dets[1])         # grouped == loop, bit for bit


@pytest.mark.parametrize("ndim", [1, 2])
def test_pipeline_convolution_then_filterbank(oracle, gpu, ndim):
    """`dspsr -F 16:B` (Config::Before): dsp::Convolution of the whole input channel, then the non-convolving filterbank on the
    dedispersed complex rows -- LoadToFold against the oracle chain (Convolution.C, then Filterbank.C with freq_res = 1)."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    o = oracle
    freq, bw, dm, period, nchan, nbin, M = 1382.0, -16.0, 30.0, 0.004, 16, 64, 32768
    tsamp = 1.0 / 32.0 if ndim == 1 else 1.0 / 16.0
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=1,
                          max_parts=2, convolve_when="before", freq_res=M)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, ndim=ndim, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    nblocks = 3
    step = cfg.parts_per_block * lt.nsamp_step
    ndat = nblocks * step + lt.nsamp_overlap
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period, ndim=ndim)
    d_raw = torch.from_numpy(raw).cuda()
    bps = 2 * ndim
    for b in range(nblocks):
        lt.process_block(d_raw[bps * b * step: bps * (b * step + step + lt.nsamp_overlap)])
    lt.finish_subint()
    lt.synchronize()
    got = lt.subints[0]["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4)
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm, ndim=ndim)
    r = o.Dedispersion()
    r.set_frequency_resolution(M)
    r.match(obs)
    assert np.abs(r.buffer - lt.response.kernel).max() <= 1.2e-7 and (r.impulse_pos, r.impulse_neg) == (lt.response.impulse_pos, lt.response.impulse_neg)
    cv = o.convolution(o.unpack_8bit(raw, obs), r.ndat, r.impulse_pos, r.impulse_neg, r.buffer, ndim == 1, dtype=np.float64)
    obs_c = o.Observation(**obs.__dict__)
    obs_c.ndim, obs_c.tsamp_us = 2, tsamp * (2 if ndim == 1 else 1)
    plan = o.filterbank_plan(obs_c, nchan, None, 1)
    fb = o.filterbank(np.ascontiguousarray(cv).view(np.float64), plan, None, dtype=np.float64)
    fobs = o.filterbank_output_observation(obs_c, plan)
    fobs.start_seconds = obs.start_seconds + r.impulse_pos / obs_c.rate
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    per_block = cfg.parts_per_block * lt.nkeep
    assert nblocks * per_block <= det.shape[2]
    for b in range(nblocks):
        o.fold(det, fobs, fcfg, ps, idat_start=b * per_block, ndat_fold=per_block)
    assert np.array_equal(lt.subints[0]["hits"], ps.hits)
    assert np.abs(got - ps.data).max() <= 1e-5 * np.abs(ps.data).max()
    assert abs(lt.out_rate - fobs.rate) <= 1e-9 * fobs.rate and abs(lt.out_start - fobs.start_seconds) <= 1e-12
    lt.close()
