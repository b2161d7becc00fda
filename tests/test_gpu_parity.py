"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (stated per BASELINE.json north_star: folded output within 1e-5 relative of the
CPU/FFTW reference; integer/index work bit-exact):
  * raw filterbank output vs the float64 oracle: max |err| <= 2e-6 * sqrt(log2 N) * rms(out) * 8
    and rms(err)/rms(out) <= 2e-6 * sqrt(log2 N)                     
  * folded profile vs float64 oracle: max |err| / max |profile| <= 1e-5 per (chan, product)
  * hits[], bin plan, fold of identical detected samples: bit-exact
"""
import math

import numpy as np
import pytest

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


def _raw(ndat, npol=2, ndim=1, nchan=1, seed=1):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.standard_normal(ndat * nchan * npol * ndim) * 24.0), -128, 127).astype(np.int8)


def _fb_case(oracle, gpu, C, M, nfilt, npart, npol=2, real=True, input_nchan=1, layout="generic", use_raw=True,
             max_parts=1, seed=3, four_pass=False):
    dspsr_amd, ctx = gpu
    o = oracle
    nchan = C * input_nchan
    N = C * M
    nfilt_pos, nfilt_neg = nfilt
    rng = np.random.default_rng(seed)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, input_nchan * N)).astype(np.complex64)
    kernel[0] = 0
    obs = o.Observation(nchan=input_nchan, npol=npol, ndim=1 if real else 2,
                        machine="CASPSR" if layout == "caspsr" else "DADA")
    plan = o.FilterbankPlan(nchan, input_nchan, C, M, N, nfilt_pos, nfilt_neg, nfilt_pos + nfilt_neg,
                            2 * N if real else N, (2 if real else 1) * (nfilt_pos + nfilt_neg) * C, 0,
                            M - nfilt_pos - nfilt_neg, float(N) * M, real)
    plan.nsamp_step = plan.nsamp_fft - plan.nsamp_overlap
    ndat = npart * plan.nsamp_step + plan.nsamp_overlap
    # (the CASPSR byte order comes in whole groups of 4 samples: 4 B pol0 | 4 B pol1)
    raw = _raw(-(-ndat // 4) * 4 if layout == "caspsr" else ndat, npol, obs.ndim, input_nchan, seed)
    scale = float(o.S8)
    unpacked = o.unpack_8bit(raw, obs)
    ref = o.filterbank(unpacked, plan, kernel, npart=npart, dtype=np.float64)

    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt_pos, nfilt_neg, input_nchan, npol, real, kernel,
                                                max_parts=max_parts, force_four_pass=four_pass)
    assert (eng.nsamp_fft, eng.nsamp_overlap, eng.nsamp_step, eng.nkeep) == \
        (plan.nsamp_fft, plan.nsamp_overlap, plan.nsamp_step, plan.nkeep)
    out = torch.zeros((nchan, npol, 2 * npart * plan.nkeep), dtype=torch.float32, device="cuda")
    if use_raw:
        d_raw = torch.from_numpy(raw).cuda()
        eng.perform_raw(d_raw, dspsr_amd.RAW_CASPSR if layout == "caspsr" else dspsr_amd.RAW_GENERIC, scale, out,
                        npart)
    else:
        d_in = torch.from_numpy(unpacked).cuda()
        eng.perform(d_in, out, npart, plan.nsamp_step * obs.ndim, 2 * plan.nkeep)
    eng.finish()
    got = out.cpu().numpy().view(np.complex64).astype(np.complex128)
    eng.close()
    err = got - ref
    rms_ref = math.sqrt(np.mean(np.abs(ref) ** 2))
    rms_err = math.sqrt(np.mean(np.abs(err) ** 2))
    tol = 2e-6 * math.sqrt(math.log2(2 * N))
    assert rms_err / rms_ref <= tol, (rms_err / rms_ref, tol)
    assert np.abs(err).max() <= 8 * tol * rms_ref, (np.abs(err).max() / rms_ref, 8 * tol)
    return got, ref


@pytest.mark.parametrize("C,M,nfilt,npart", [
    (8, 64, (5, 7), 3),          # tiny: one workgroup per pass
    (4, 16, (1, 2), 2),          # below one wave per workgroup
    (16, 256, (20, 21), 2),
    (32, 512, (40, 30), 3),      # radix 16*16*2 / rows 64
    (64, 1024, (100, 101), 2),
    (2, 2048, (100, 50), 2),     # few channels, long backward FFT
    (512, 32, (3, 4), 2),        # many channels, short backward FFT
])
def test_filterbank_real_dualpol_raw(oracle, gpu, C, M, nfilt, npart):
    _fb_case(oracle, gpu, C, M, nfilt, npart)


def test_filterbank_cfg2_size(oracle, gpu):
    # BASELINE cfg2 geometry: -F 256:D -x 4096 (N = 2^20), nfilt 953/956
    _fb_case(oracle, gpu, 256, 4096, (953, 956), 2, max_parts=2)


def test_filterbank_target_size(oracle, gpu):
    # headline geometry: -F 1024:D -x 4096 (N = 2^22), DM 1000 -> 422/422
    _fb_case(oracle, gpu, 1024, 4096, (422, 422), 1)


def test_filterbank_float_input_equals_raw(oracle, gpu):
    a, _ = _fb_case(oracle, gpu, 16, 256, (20, 21), 2, use_raw=True)
    b, _ = _fb_case(oracle, gpu, 16, 256, (20, 21), 2, use_raw=False)
    # same arithmetic after the (int8+0.5)*scale conversion, except that in the fused path the compiler may
    # contract the conversion multiply into the first butterfly add (one rounding less)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()


def test_filterbank_caspsr_layout(oracle, gpu):
    _fb_case(oracle, gpu, 16, 256, (20, 21), 2, layout="caspsr")


def test_filterbank_subband_shard_size(oracle, gpu):
    # BASELINE cfg 4 geometry: one 50 MHz complex dual-pol sub-band, -F 512:D (N = 2^18, freq_res 512), 27/27 at DM 1000;
    # 8-bit complex input takes the per-polarisation regroup pass and the one-word-per-pair loads of pass 1
    _fb_case(oracle, gpu, 512, 512, (27, 27), 2, npol=2, real=False, max_parts=2)
    _fb_case(oracle, gpu, 512, 512, (27, 27), 3, npol=2, real=False, max_parts=2, use_raw=False)


def test_filterbank_single_pol(oracle, gpu):
    _fb_case(oracle, gpu, 16, 256, (20, 21), 2, npol=1)


@pytest.mark.parametrize("input_nchan,npol", [(1, 2), (4, 2), (2, 1)])
def test_filterbank_complex_input(oracle, gpu, input_nchan, npol):
    _fb_case(oracle, gpu, 32, 128, (9, 10), 2, npol=npol, real=False, input_nchan=input_nchan)
    _fb_case(oracle, gpu, 32, 128, (9, 10), 2, npol=npol, real=False, input_nchan=input_nchan, use_raw=False)


def test_filterbank_batched_parts_identical(oracle, gpu):
    a, _ = _fb_case(oracle, gpu, 16, 256, (20, 21), 5, max_parts=1)
    b, _ = _fb_case(oracle, gpu, 16, 256, (20, 21), 5, max_parts=4)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("C,M,nfilt,npart", [
    (8, 64, (5, 7), 3),
    (16, 256, (20, 21), 2),
    (64, 1024, (100, 101), 2),
    (2, 2048, (100, 50), 2),
    (512, 32, (3, 4), 2),
])
def test_filterbank_four_pass_forced(oracle, gpu, C, M, nfilt, npart):
    """force_four_pass: the two-pass inverse (k_inv_a + k_inv_b) also where three passes would do."""
    _fb_case(oracle, gpu, C, M, nfilt, npart, four_pass=True)


def test_filterbank_four_pass_other_inputs(oracle, gpu):
    _fb_case(oracle, gpu, 16, 256, (20, 21), 2, layout="caspsr", four_pass=True)
    _fb_case(oracle, gpu, 16, 256, (20, 21), 2, npol=1, four_pass=True)
    _fb_case(oracle, gpu, 16, 256, (20, 21), 3, use_raw=False, max_parts=2, four_pass=True)
    for input_nchan, npol in [(1, 2), (4, 2), (2, 1)]:
        _fb_case(oracle, gpu, 32, 128, (9, 10), 2, npol=npol, real=False, input_nchan=input_nchan, four_pass=True)


@pytest.mark.parametrize("C,M,nfilt,real", [
    (4, 16384, (900, 1100), True),       # freq_res beyond one workgroup tile: BASELINE cfg 1 (-F 64:D) class
    (2, 65536, (7226, 7341), True),
    (8, 16384, (900, 1100), False),
])
def test_filterbank_large_freq_res(oracle, gpu, C, M, nfilt, real):
    _fb_case(oracle, gpu, C, M, nfilt, 2, real=real, max_parts=2)


def test_filterbank_cfg1_size(oracle, gpu):
    # BASELINE cfg 1: -F 64:D on the header.dada band with the minimum response length 16384 (N = 2^20)
    _fb_case(oracle, gpu, 64, 16384, (7226, 7341), 1)


@pytest.mark.parametrize("M,nfilt,real,input_nchan,npol", [
    (4096, (300, 320), False, 8, 2),     # dsp::Convolution after a filterbank: nchan_subband = 1 per input channel
    (1024, (100, 90), False, 3, 1),
    (32768, (3000, 2000), True, 1, 2),   # single-channel coherent dedispersion of real dual-pol data
    (131072, (10000, 11000), False, 1, 2),
])
def test_convolution_nchan_subband_1(oracle, gpu, M, nfilt, real, input_nchan, npol):
    _fb_case(oracle, gpu, 1, M, nfilt, 2, npol=npol, real=real, input_nchan=input_nchan, max_parts=2)


@pytest.mark.parametrize("ndat,nfilt,real,nchan,npol", [(2048, (200, 150), False, 16, 2), (8192, (700, 800), True, 2, 2)])
def test_convolution_engine(oracle, gpu, ndat, nfilt, real, nchan, npol):
    """HIP ConvolutionEngine (dsp::Convolution::Engine mirror) against the oracle's restatement of Convolution.C."""
    dspsr_amd, ctx = gpu
    o = oracle
    npart = 3
    rng = np.random.default_rng(11)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, nchan * ndat)).astype(np.complex64)
    ndim = 1 if real else 2
    nfilt_tot = sum(nfilt)
    nsamp_fft, nsamp_overlap = (2 * ndat, 2 * nfilt_tot) if real else (ndat, nfilt_tot)
    nsamp_step = nsamp_fft - nsamp_overlap
    x = rng.standard_normal((nchan, npol, (npart * nsamp_step + nsamp_overlap) * ndim)).astype(np.float32)
    ref = o.convolution(x, ndat, nfilt[0], nfilt[1], kernel, real, npart=npart, dtype=np.float64)
    eng = dspsr_amd.ConvolutionEngine(ctx).prepare(ndat, nfilt[0], nfilt[1], nchan, npol, real, kernel, max_parts=2)
    assert (eng.nsamp_fft, eng.nsamp_overlap, eng.nsamp_step) == (nsamp_fft, nsamp_overlap, nsamp_step)
    ngood = ndat - nfilt_tot
    out = torch.zeros((nchan, npol, 2 * npart * ngood), dtype=torch.float32, device="cuda")
    eng.perform(torch.from_numpy(x).cuda(), out, npart, nsamp_step * ndim, 2 * ngood)
    eng.finish()
    got = out.cpu().numpy().view(np.complex64).astype(np.complex128)
    eng.close()
    rms_ref = math.sqrt(np.mean(np.abs(ref) ** 2))
    tol = 2e-6 * math.sqrt(math.log2(2 * ndat))
    assert math.sqrt(np.mean(np.abs(got - ref) ** 2)) / rms_ref <= tol
    assert np.abs(got - ref).max() <= 8 * tol * rms_ref


@pytest.mark.parametrize("npol,C,M,nfilt", [(2, 8, 256, (30, 31)), (1, 1, 4096, (300, 301))])
def test_filterbank_uwb_16bit_input(oracle, gpu, npol, C, M, nfilt):
    """16-bit offset-binary complex input (UWB layout) decoded in pass 1 vs the oracle's unpacker + filterbank."""
    dspsr_amd, ctx = gpu
    o = oracle
    N = C * M
    ovl = sum(nfilt) * C
    step = N - ovl
    npart = 3
    ndat = -(-(npart * step + ovl) // 2048) * 2048
    rng = np.random.default_rng(17)
    raw = np.clip(np.rint(rng.standard_normal(ndat * npol * 2) * 3000.0), -32768, 32767).astype(np.int16)
    raw = (raw.view(np.uint16) ^ np.uint16(0x8000))                       # offset binary
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    unpacked = o.unpack_uwb16(raw, npol)
    plan = o.FilterbankPlan(C, 1, C, M, N, nfilt[0], nfilt[1], sum(nfilt), N, ovl, step, M - sum(nfilt), float(N) * M, False)
    ref = o.filterbank(unpacked, plan, kernel, npart=npart, dtype=np.float64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, npol, False, kernel, max_parts=2)
    out = torch.zeros((C, npol, 2 * npart * plan.nkeep), dtype=torch.float32, device="cuda")
    eng.perform_raw(torch.from_numpy(raw.view(np.int8)).cuda(), dspsr_amd.RAW_UWB16, 1.0, out, npart)
    eng.finish()
    got = out.cpu().numpy().view(np.complex64).astype(np.complex128)
    eng.close()
    rms_ref = math.sqrt(np.mean(np.abs(ref) ** 2))
    tol = 2e-6 * math.sqrt(math.log2(2 * N))
    assert math.sqrt(np.mean(np.abs(got - ref) ** 2)) / rms_ref <= tol
    assert np.abs(got - ref).max() <= 8 * tol * rms_ref


def test_copy_data_fpt(gpu):
    """dsp::TimeSeries::Engine::copy_data_fpt twin: row copies with independent strides (bit-exact)."""
    dspsr_amd, ctx = gpu
    nchan, npol, n = 5, 2, 1003
    src = torch.randn((nchan, npol, 1500), device="cuda")
    dst = torch.zeros((nchan, npol, 1200), device="cuda")
    for off in (0, 3, 4):                                   # aligned and unaligned start samples
        dst.zero_()
        lib = dspsr_amd.lib
        rc = lib.dspsr_amd_copy_fpt(ctx.handle, dst.data_ptr(), dst.stride(0), dst.stride(1),
                                    src.data_ptr() + 4 * off, src.stride(0), src.stride(1), nchan, npol, n)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(dst[:, :, :n], src[:, :, off:off + n]) and float(dst[:, :, n:].abs().max()) == 0.0


def test_edge_cases_empty_and_ragged(oracle, gpu):
    """npart = 0 is a no-op, an empty fold plan leaves the profile untouched, a ragged last block (fewer parts
    than the block size) folds like the oracle, and perform_fold on a four-pass geometry takes the unfused chain."""
    dspsr_amd, ctx = gpu
    o = oracle
    C, M, nfilt = 8, 64, (5, 7)
    kernel = np.ones(C * M, np.complex64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, True, kernel, max_parts=2)
    out = torch.full((C, 2, 8), 7.0, dtype=torch.float32, device="cuda")
    raw = torch.zeros(4096, dtype=torch.int8, device="cuda")
    eng.perform_raw(raw, dspsr_amd.RAW_GENERIC, 1.0, out, 0)                       # zero parts
    eng.finish()
    assert float(out.min()) == 7.0 and float(out.max()) == 7.0
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(C, 1, 4, 16)
    fold.set_nbin(16)
    fold.set_ndat(0, 0)
    fold.fold(torch.zeros((C, 1, 16), device="cuda"))                             # empty plan
    eng.perform_fold(fold, 0, dspsr_amd.COHERENCE, raw=raw, scale=1.0)             # zero parts, empty plan
    assert float(np.abs(fold.synch()).max()) == 0.0
    eng.close()
    # perform_fold on a four-pass geometry with NARROW phase bins (4.9 samples each): mode 3 applies to wide bins only, so
    # the library runs Detection and the exact Fold itself -- same sums
    big = dspsr_amd.FilterbankEngine(ctx).setup(2, 16384, 100, 100, 1, 2, True, None, max_parts=1)
    assert big.fold_is_fused() == 3
    nk = 16384 - 200
    braw = torch.from_numpy(_raw(1 << 16, seed=9)).cuda()
    f2, f3 = dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)
    h2, h3 = np.zeros(16, np.uint32), np.zeros(16, np.uint32)
    for f, h in ((f2, h2), (f3, h3)):
        f.set_shape(2, 1, 4, 16)
        f.set_nbin(16)
        f.set_ndat(nk, 0)
        f.set_bins(0.1, 1.0 / 77.7, nk, 0, h)
    big.perform_fold(f2, 1, dspsr_amd.COHERENCE, raw=braw, scale=1.0)
    bdet = torch.zeros((2, 1, 4 * nk), dtype=torch.float32, device="cuda")
    big.perform_detect(bdet, 1, dspsr_amd.COHERENCE, 4, raw=braw, scale=1.0)
    f3.fold(bdet)
    assert np.array_equal(f2.synch(), f3.synch()) and np.abs(f2.synch()).max() > 0
    big.close()
    f2.close()
    f3.close()
    fold.close()
    # ragged block through the pipeline driver: 3 parts, then 1 part
    from dspsr_amd import pipeline, synth
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4,
                          parts_per_block=3, max_parts=2)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    step = lt.nsamp_step
    rawh = synth.voltages(4 * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)
    d_raw = torch.from_numpy(rawh).cuda()
    lt.process_block(d_raw[: 2 * (3 * step + lt.nsamp_overlap)], npart=3)
    lt.process_block(d_raw[2 * 3 * step:], npart=1)
    lt.finish_subint()
    lt.synchronize()
    got = lt.subints[0]["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4)
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm)
    resp = o.Dedispersion().match(obs, nchan)
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(o.unpack_8bit(rawh, obs), plan, lt.response.kernel, dtype=np.float64)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    o.fold(det, fobs, fcfg, ps, idat_start=0, ndat_fold=3 * plan.nkeep)
    o.fold(det, fobs, fcfg, ps, idat_start=3 * plan.nkeep, ndat_fold=plan.nkeep)
    assert np.array_equal(lt.subints[0]["hits"], ps.hits)
    assert np.abs(got - ps.data).max() <= 1e-5 * np.abs(ps.data).max()
    lt.close()


def test_pipeline_folds_with_tempo2_predictor(oracle, gpu):
    """dspsr -P <ChebyModelSet>: phase and period from a TEMPO2 predictor at each block's first sample (Fold.C:943-958 through
    the duck-typed predictor interface) -- the pipeline with pipeline.ChebyPredictor against the oracle's fold with the
    oracle's own evaluator: hits identical, profile within 1e-5."""
    import cheby_fixture as cf
    from dspsr_amd import pipeline, synth
    o = oracle
    text = cf.cheby_text()
    freq, bw, tsamp, dm, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=0.0, ndim=4, parts_per_block=2, max_parts=2)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, polyco=pipeline.ChebyPredictor(text, freq))
    step = 2 * lt.nsamp_step
    rawh = synth.voltages(2 * step + lt.nsamp_overlap, freq, bw, tsamp, dm, 0.004)
    d_raw = torch.from_numpy(rawh).cuda()
    for b in range(2):
        lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.finish_subint()
    lt.synchronize()
    got = lt.subints[0]["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4)
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm)
    resp = o.Dedispersion().match(obs, nchan)
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(o.unpack_8bit(rawh, obs), plan, lt.response.kernel, dtype=np.float64)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    fcfg = o.FoldConfig(nbin=nbin, polyco=o.ChebyPredictor(text, freq))
    for b in range(2):
        o.fold(det, fobs, fcfg, ps, idat_start=2 * b * plan.nkeep, ndat_fold=2 * plan.nkeep)
    assert np.array_equal(lt.subints[0]["hits"], ps.hits) and int(ps.hits.sum()) == 4 * plan.nkeep
    assert np.abs(got - ps.data).max() <= 1e-5 * np.abs(ps.data).max()
    lt.close()


def test_filterbank_errors(gpu):
    dspsr_amd, ctx = gpu
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.FilterbankEngine(ctx).setup(8, 8 * 131, 1, 1)       # freq_res is not 2^k times an odd factor <= 127
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.FilterbankEngine(ctx).setup(8, 64, 40, 30)          # nfilt_tot >= freq_res
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.FilterbankEngine(ctx).setup(8, 64, 1, 1, kernel=np.zeros(5, np.complex64))


@pytest.mark.parametrize("state", ["Coherence", "Stokes"])
@pytest.mark.parametrize("ndim", [1, 2, 4])
def test_detection_engine(oracle, gpu, state, ndim):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(5)
    nchan, ndat = 6, 1000
    x = rng.standard_normal((nchan, 2, 2 * ndat)).astype(np.float32)
    fb = x.view(np.complex64)
    prod = oracle.detect_products(fb, state)                    # float32 arithmetic as the reference
    want = oracle.detect_layout(prod, ndim)
    npol_out = 4 // ndim
    out = torch.zeros((nchan, npol_out, ndat * ndim), dtype=torch.float32, device="cuda")
    st = dspsr_amd.COHERENCE if state == "Coherence" else dspsr_amd.STOKES
    dspsr_amd.DetectionEngine(ctx).polarimetry(ndim, torch.from_numpy(x).cuda(), out, st)
    got = out.cpu().numpy().reshape(want.shape if ndim > 1 else (nchan, 4, ndat))
    # same float32 operations; allow 1 ulp for fused multiply-add contraction on the device
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=3e-7 * np.abs(want).max())


def test_detection_inplace_ndim2(oracle, gpu):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 2, 512)).astype(np.float32)
    want = oracle.detect_layout(oracle.detect_products(x.view(np.complex64), "Coherence"), 2)
    d = torch.from_numpy(x).cuda()
    dspsr_amd.DetectionEngine(ctx).polarimetry(2, d, d)          # LoadToFold1.C:545-546 in-place use
    np.testing.assert_allclose(d.cpu().numpy().reshape(want.shape), want, rtol=3e-7, atol=1e-6)


def test_square_law(oracle, gpu):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 2, 600)).astype(np.float32)
    fb = x.view(np.complex64)
    for intensity in (False, True):
        want = oracle.square_law(fb, "Intensity" if intensity else "PPQQ")
        out = torch.zeros((5, 1 if intensity else 2, 300), dtype=torch.float32, device="cuda")
        dspsr_amd.DetectionEngine(ctx).square_law(torch.from_numpy(x).cuda(), out, intensity)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=3e-7, atol=1e-6)


@pytest.mark.parametrize("ndim,npol", [(4, 1), (2, 2), (1, 4)])
@pytest.mark.parametrize("nbin", [64, 1024, 100])
def test_fold_engine_bit_exact(oracle, gpu, ndim, npol, nbin):
    """Fold of identical detected samples is bit-identical to the CPU loop (Fold.C:835-891)."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(8)
    nchan, ndat = 5, 7000
    det = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32) ** 2
    phi, pps = 0.37, 1.0 / 333.3
    idat_start, ndat_fold = 11, ndat - 50
    eng = dspsr_amd.FoldEngine(ctx)
    eng.set_shape(nchan, npol, ndim, nbin)
    d = torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda()
    want = np.zeros((nchan, npol, nbin, ndim), np.float32)
    hits_total = np.zeros(nbin, np.uint32)
    for rep in range(2):                                   # accumulate twice: += semantics
        hits = np.zeros(nbin, np.uint32)
        eng.set_nbin(nbin)
        eng.set_ndat(ndat_fold, idat_start)
        n = eng.set_bins(phi, pps, ndat_fold, idat_start, hits)
        assert n == ndat_fold and eng.get_ndat_folded() == ndat_fold
        eng.fold(d)
        plan = oracle.fold_binplan(phi, pps, nbin, ndat_fold)
        assert np.array_equal(hits, np.bincount(plan, minlength=nbin).astype(np.uint32))
        hits_total += hits
        for i in range(ndat_fold):
            want[:, :, plan[i], :] += det[:, :, idat_start + i, :]
    got = eng.synch()
    assert np.array_equal(got, want)
    eng.zero()
    assert not eng.synch().any()
    eng.close()


@pytest.mark.parametrize("ndim,npol", [(2, 2), (1, 4)])
@pytest.mark.parametrize("pps", [1.0 / 7.7, 1.0 / 90000.0])
def test_fold_planes_of_a_channel_together_bit_exact(oracle, gpu, ndim, npol, pps):
    """With enough channels to fill the chip the 4/ndim planes of a channel are folded by ONE workgroup (the walk through
    the bin plan serves all planes; the reference's fold benchmark, Benchmark/fold.csh, is this shape: 1024 channels x 4
    products).  Short runs: still bit-identical to the CPU loop; runs of thousands of samples: the re-associated long-run
    fold, equal to rounding and identical to folding the planes one by one."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(18)
    nchan, ndat, nbin = 600, 9000, 200
    det = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32) ** 2
    phi, idat_start, ndat_fold = 0.61, 4, ndat - 9
    d = torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda()
    eng = dspsr_amd.FoldEngine(ctx)
    eng.set_shape(nchan, npol, ndim, nbin)
    hits = np.zeros(nbin, np.uint32)
    eng.set_nbin(nbin)
    eng.set_ndat(ndat_fold, idat_start)
    eng.set_bins(phi, pps, ndat_fold, idat_start, hits)
    eng.fold(d)
    got = eng.synch()
    plan = oracle.fold_binplan(phi, pps, nbin, ndat_fold)
    assert np.array_equal(hits, np.bincount(plan, minlength=nbin).astype(np.uint32))
    want = np.zeros((nchan, npol, nbin, ndim), np.float32)
    for i in range(ndat_fold):
        want[:, :, plan[i], :] += det[:, :, idat_start + i, :]
    if pps > 1e-3:
        assert np.array_equal(got, want)
    else:
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
        # the planes one by one (few channels -> one row per workgroup): the same re-associated sums
        one = dspsr_amd.FoldEngine(ctx)
        one.set_shape(4, npol, ndim, nbin)
        one.set_nbin(nbin)
        one.set_ndat(ndat_fold, idat_start)
        one.set_bins(phi, pps, ndat_fold, idat_start, np.zeros(nbin, np.uint32))
        one.fold(d[:4])
        got4 = one.synch()
        one.close()
        assert np.abs(got[:4] - got4).max() <= 2e-6 * np.abs(want).max()
    eng.close()


def test_fold_set_bin_equals_set_bins(gpu):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(9)
    det = rng.standard_normal((2, 1, 3000 * 4)).astype(np.float32)
    d = torch.from_numpy(det).cuda()
    res = []
    for mode in (0, 1):
        eng = dspsr_amd.FoldEngine(ctx)
        eng.set_shape(2, 1, 4, 128)
        eng.set_nbin(128)
        eng.set_ndat(3000, 0)
        phi, pps = 0.9, 0.0123
        if mode == 0:
            eng.set_bins(phi, pps, 3000, 0)
        else:                                             # the per-sample host loop of Fold.C:744-787
            for i in range(3000):
                phi -= math.floor(phi)
                eng.set_bin(i, phi * 128.0, pps * 128.0)
                phi += pps
        eng.fold(d)
        res.append(eng.synch())
        eng.close()
    assert np.array_equal(res[0], res[1])


def test_end_to_end_folded_profile(oracle, gpu):
    """raw bytes -> fused filterbank+detect -> fold, against the float64 oracle pipeline (<= 1e-5 relative)."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import synth
    o = oracle
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, npol=2, ndim=1)
    resp = dspsr_amd.Dedispersion(freq, bw, dm)
    resp.match(nchan)
    oresp = o.Dedispersion()
    obs.dispersion_measure = dm
    oresp.match(obs, nchan)
    assert (resp.impulse_pos, resp.impulse_neg, resp.ndat) == (oresp.impulse_pos, oresp.impulse_neg, oresp.ndat)
    plan = o.filterbank_plan(obs, nchan, oresp)
    npart = 6
    ndat = npart * plan.nsamp_step + plan.nsamp_overlap
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period)
    # oracle (float64 arithmetic, product's own kernel so only the device path differs)
    unpacked = o.unpack_8bit(raw, obs)
    fb = o.filterbank(unpacked, plan, resp.kernel, dtype=np.float64)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
    cfg = o.FoldConfig(nbin=nbin, folding_period=period)
    o.fold(det, fobs, cfg, ps)
    # device
    eng = dspsr_amd.FilterbankEngine(ctx).setup(plan.nchan_subband, plan.freq_res, plan.nfilt_pos, plan.nfilt_neg, 1,
                                                2, True, resp.kernel, max_parts=4)
    d_det = torch.zeros((nchan, 1, 4 * npart * plan.nkeep), dtype=torch.float32, device="cuda")
    eng.perform_detect(d_det, npart, dspsr_amd.COHERENCE, 4, raw=torch.from_numpy(raw).cuda(),
                       scale=float(o.S8))
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(nchan, 1, 4, nbin)
    fold.set_nbin(nbin)
    ndat_out = npart * plan.nkeep
    phi, pfold = o.fold_phase(cfg, fobs, fobs.start_seconds + 0.5 / fobs.rate)
    hits = np.zeros(nbin, np.uint32)
    fold.set_ndat(ndat_out, 0)
    fold.set_bins(phi, (1.0 / fobs.rate) / pfold, ndat_out, 0, hits)
    fold.fold(d_det)
    got = fold.synch().astype(np.float64)
    assert np.array_equal(hits, ps.hits)
    for c in range(nchan):
        for k in range(4):
            ref = ps.data[c, 0, :, k]
            err = np.abs(got[c, 0, :, k] - ref).max() / np.abs(ps.data[c, 0, :, :2]).max()
            assert err <= 1e-5, (c, k, err)
    # the dedispersed pulse must be sharp: on-pulse bins well above the off-pulse mean
    # (per channel: inter-channel delays are not removed by the filterbank, Dedispersion.C:524-545 needs -K)
    prof = got[0, 0, :, 0] / hits
    assert prof.max() > 1.5 * np.median(prof)
    eng.close()
    fold.close()


@pytest.mark.parametrize("C,M,nfilt,nbin,period_samples,state", [
    (16, 256, (20, 21), 64, 97.3, "Coherence"),       # ~1.5 samples per bin: every bin touched in every part
    (8, 1024, (100, 101), 1024, 34766.4, "Coherence"),  # long period: ~34 samples per bin, few bins per part
    (64, 128, (9, 10), 100, 1234.5, "Stokes"),          # nbin not a power of two
    (2, 2048, (100, 50), 1500, 333.3, "Coherence"),     # nbin beyond the workgroup size
])
@pytest.mark.parametrize("force_fused", [True, False])
def test_fused_fold_bit_identical(oracle, gpu, C, M, nfilt, nbin, period_samples, state, force_fused):
    """perform_fold (fold inside the last filterbank pass) == perform_detect + FoldEngine.fold, bit for bit,
    over several calls and launch groups (accumulators re-loaded from the device profile).  These geometries have
    fewer channel tiles than the chip has compute units, where the library would not fuse by itself:
    fused_fold=FUSED_ALWAYS forces the fused kernel, the default takes the internal Detection + Fold chain."""
    dspsr_amd, ctx = gpu
    o = oracle
    N = C * M
    nkeep = M - sum(nfilt)
    step, ovl = 2 * (N - sum(nfilt) * C), 2 * sum(nfilt) * C
    npart, ncall = 5, 3
    rng = np.random.default_rng(23)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    st = dspsr_amd.STOKES if state == "Stokes" else dspsr_amd.COHERENCE
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, True, kernel, max_parts=2,
                                                fused_fold=dspsr_amd.FUSED_ALWAYS if force_fused else dspsr_amd.FUSED_AUTO)
    assert eng.fold_is_fused() == force_fused
    folds = [dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)]
    hits = [np.zeros(nbin, np.uint32), np.zeros(nbin, np.uint32)]
    for f in folds:
        f.set_shape(C, 1, 4, nbin)
    det = torch.zeros((C, 1, 4 * npart * nkeep), dtype=torch.float32, device="cuda")
    pps = 1.0 / period_samples
    for call in range(ncall):
        raw = torch.from_numpy(_raw(npart * step + ovl, seed=100 + call)).cuda()
        ndat = npart * nkeep
        phi = (0.37 + call * ndat * pps) % 1.0
        for k, f in enumerate(folds):
            f.set_nbin(nbin)
            f.set_ndat(ndat, 0)
            f.set_bins(phi, pps, ndat, 0, hits[k])
        eng.perform_detect(det, npart, st, 4, raw=raw, scale=float(o.S8))
        folds[0].fold(det)
        eng.perform_fold(folds[1], npart, st, raw=raw, scale=float(o.S8))
    a, b = folds[0].synch(), folds[1].synch()
    assert np.array_equal(hits[0], hits[1])
    assert np.abs(a).max() > 0 and np.array_equal(a, b)
    eng.close()
    for f in folds:
        f.close()


def test_pipeline_fused_equals_unfused(oracle, gpu):
    """LoadToFold with and without the fused fold over blocks with sub-integration boundaries: identical dumps."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    dumps = []
    for fused in (True, False):
        cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4,
                              parts_per_block=3, max_parts=2, subint_seconds=0.004, fused_fold=fused,
                              force_fused=fused)      # small geometry: fuse although it does not fill the chip
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        assert lt.fused_fold == fused
        nblocks = 5
        step = cfg.parts_per_block * lt.nsamp_step
        raw = synth.voltages(nblocks * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)
        d_raw = torch.from_numpy(raw).cuda()
        for b in range(nblocks):
            lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
        lt.finish_subint()
        lt.synchronize()
        dumps.append([(s["hits"].copy(), s["profile_dev"].cpu().numpy(), s["integration_length"]) for s in lt.subints])
        lt.close()
    assert len(dumps[0]) == len(dumps[1]) >= 2
    for (h0, p0, t0), (h1, p1, t1) in zip(*dumps):
        assert np.array_equal(h0, h1) and np.array_equal(p0, p1) and t0 == t1


def test_pipeline_subintegrations(oracle, gpu):
    """LoadToFold driver with -L sub-integrations over several blocks (overlap carried between blocks) against
    the oracle folding the same pieces: identical hits per sub-integration, profiles <= 1e-5."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    o = oracle
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4,
                          parts_per_block=3, max_parts=2, subint_seconds=0.002)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    nblocks = 4
    step = cfg.parts_per_block * lt.nsamp_step
    ndat = nblocks * step + lt.nsamp_overlap
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period)
    d_raw = torch.from_numpy(raw).cuda()
    for b in range(nblocks):
        lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.synchronize()
    got = [(s["hits"], s["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4)) for s in lt.subints]
    assert len(got) >= 2
    # oracle: whole stream in one go, then fold division by division
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm)
    resp = o.Dedispersion().match(obs, nchan)
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(o.unpack_8bit(raw, obs), plan, lt.response.kernel, dtype=np.float64)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    block_out = cfg.parts_per_block * plan.nkeep
    for isub, (hits, prof) in enumerate(got):
        ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
        lo, hi = o.subint_sample_bounds(fobs, cfg.subint_seconds, isub)
        # the product folds piecewise (block and division boundaries restart the phase recurrence, Fold.C:650-657)
        pos = lo
        while pos < hi:
            stop = min(hi, (pos // block_out + 1) * block_out)
            o.fold(det, fobs, fcfg, ps, idat_start=pos, ndat_fold=stop - pos)
            pos = stop
        assert np.array_equal(hits, ps.hits), isub
        assert np.abs(prof - ps.data).max() <= 1e-5 * np.abs(ps.data).max(), isub
    lt.close()


@pytest.mark.parametrize("turns", [1.0, 2.0, 0.5])             # 0.5: phase-resolved divisions (half turns)
def test_pipeline_turns_mode_subintegrations(oracle, gpu, turns):
    """dspsr -s / -turns N: sub-integrations of N pulse periods (TimeDivide.C:360-500) through the LoadToFold driver over
    several blocks, against the oracle folding the same divisions: the leading partial turn is dropped, every division
    holds N turns of samples, hits identical, profiles <= 1e-5."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    o = oracle
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4,
                          parts_per_block=3, max_parts=2, subint_turns=turns)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    nblocks = 8
    step = cfg.parts_per_block * lt.nsamp_step
    raw = synth.voltages(nblocks * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)
    d_raw = torch.from_numpy(raw).cuda()
    for b in range(nblocks):
        lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.synchronize()
    got = [(s["hits"], s["profile_dev"].cpu().numpy().reshape(nchan, 1, nbin, 4), s["ndat_total"]) for s in lt.subints]
    assert len(got) >= 2
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm)
    resp = o.Dedispersion().match(obs, nchan)
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(o.unpack_8bit(raw, obs), plan, lt.response.kernel, dtype=np.float64)
    det = o.detect_layout(o.detect_products(fb, "Coherence"), 4)
    fobs = o.filterbank_output_observation(obs, plan)
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    block_out = cfg.parts_per_block * plan.nkeep
    first = o.subint_turns_sample_bounds(fcfg, fobs, turns, 0)[0]
    assert 0 <= first <= round(min(turns, 1.0) * period * fobs.rate) + 1     # the leading partial turn (division) is not folded
    for isub, (hits, prof, ndat_total) in enumerate(got):
        ps = o.PhaseSeries(nchan, 1, 4, nbin, data=np.zeros((nchan, 1, nbin, 4), np.float64))
        lo, hi = o.subint_turns_sample_bounds(fcfg, fobs, turns, isub)
        assert abs((hi - lo) - turns * period * fobs.rate) <= 1 and ndat_total == hi - lo
        pos = lo
        while pos < hi:                                             # block boundaries restart the phase recurrence
            stop = min(hi, (pos // block_out + 1) * block_out)
            o.fold(det, fobs, fcfg, ps, idat_start=pos, ndat_fold=stop - pos)
            pos = stop
        assert np.array_equal(hits, ps.hits), isub
        assert np.abs(prof - ps.data).max() <= 1e-5 * np.abs(ps.data).max(), isub
    lt.close()


@pytest.mark.parametrize("nchan,tscrunch,pscrunch,npart", [(4096, 16, True, 64), (4096, 16, False, 35), (256, 4, True, 130),
                                                           (64, 1, False, 300), (1024, 32, True, 96), (16, 2, False, 1024),
                                                           (8192, 4, True, 24), (8192, 1, False, 5)])
def test_tfp_filterbank_search_mode(oracle, gpu, nchan, tscrunch, pscrunch, npart):
    """digifil front end: TFPFilterbank + pscrunch + TScrunch fused, against the float32 oracle
    (same operation order); tolerance 2e-5 of the rms power (FFT rounding differs from pocketfft)."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(12)
    ndat = npart * 2 * nchan
    raw = np.clip(np.rint(rng.standard_normal(ndat * 2) * 24.0), -128, 127).astype(np.int8)
    obs = oracle.Observation()
    want = oracle.tscrunch_tfp(oracle.tfp_filterbank(oracle.unpack_8bit(raw, obs), nchan, pscrunch), tscrunch)
    nout = npart // tscrunch
    out = torch.zeros((nout, nchan, 1 if pscrunch else 2), dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, torch.from_numpy(raw).cuda(), nchan, npart, out, pscrunch, tscrunch,
                             scale=float(oracle.S8))
    got = out.cpu().numpy()
    assert got.shape == want.shape
    rms = np.sqrt(np.mean(want.astype(np.float64) ** 2))
    assert np.abs(got - want).max() <= 2e-5 * rms * 8
    assert np.sqrt(np.mean((got - want).astype(np.float64) ** 2)) <= 2e-6 * rms


@pytest.mark.parametrize("pscrunch", [True, False])
@pytest.mark.parametrize("layout_name,tscrunch,npart", [("generic", 16, 80), ("caspsr", 2, 14), ("generic", 6, 37)])
def test_tfp_4096_register_split_equals_generic_kernel(gpu, pscrunch, layout_name, tscrunch, npart):
    """nchan = 4096 with an even tscrunch takes k_tfp4k (Hermitian split, powers, pol and time sums on the last stage's registers,
    DPP adds inside the quad); tscrunch = 1 takes the generic k_tfp (staged transform).  Summing the generic kernel's per-part
    powers in time order (float32, sequentially: TScrunch.C:193-200) must reproduce k_tfp4k BIT FOR BIT (the same twiddles, the same
    operations on samples decoded at half scale -- a power of two commutes with every rounding), for both raw layouts, with and
    without pscrunch, and with a ragged last output sample dropped.  An unaligned block (generic kernel for any tscrunch) gives
    the same bits."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import _lib
    rng = np.random.default_rng(77)
    nchan = 4096
    layout = _lib.RAW_CASPSR if layout_name == "caspsr" else _lib.RAW_GENERIC
    raw = torch.from_numpy(np.clip(np.rint(rng.standard_normal(npart * 2 * nchan * 2) * 24.0), -128, 127).astype(np.int8)).cuda()
    npol = 1 if pscrunch else 2
    nout = npart // tscrunch
    one = torch.zeros((npart, nchan, npol), dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, raw, nchan, npart, one, pscrunch, 1, layout=layout, scale=0.0123)
    want = one[0:nout * tscrunch:tscrunch].clone()
    for i in range(1, tscrunch):
        want = want + one[i:nout * tscrunch:tscrunch]
    got = torch.full((nout, nchan, npol), -1.0, dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, raw, nchan, npart, got, pscrunch, tscrunch, layout=layout, scale=0.0123)
    assert torch.equal(got, want) and float(got.min()) >= 0.0 and float(got.max()) > 0.0
    if layout_name == "generic":
        shifted = torch.zeros(raw.numel() + 2, dtype=torch.int8, device="cuda")
        shifted[2:] = raw                                               # 2-byte aligned only: no 16-byte pieces, generic kernel
        got2 = torch.zeros_like(got)
        dspsr_amd.tfp_filterbank(ctx, shifted[2:], nchan, npart, got2, pscrunch, tscrunch, layout=layout, scale=0.0123)
        assert torch.equal(got2, want)


@pytest.mark.parametrize("pscrunch", [True, False])
@pytest.mark.parametrize("nchan,layout_name,tscrunch,npart", [
    (2048, "generic", 16, 100), (2048, "caspsr", 4, 30), (2048, "generic", 12, 50),
    (1024, "generic", 16, 200), (1024, "caspsr", 8, 70), (1024, "generic", 24, 100),
    (8192, "generic", 16, 40), (8192, "caspsr", 1, 5), (8192, "generic", 3, 11),
    (512, "generic", 16, 300), (512, "caspsr", 32, 100)])
def test_tfp_register_split_other_channel_counts_equal_the_generic_kernel(gpu, nchan, pscrunch, layout_name, tscrunch, npart):
    """k_tfpm (nchan = 512 / 1024 / 2048 / 8192, tscrunch a multiple of the 16 / 8 / 4 / 1 parts of a tile): the last stage's registers hold
    several (part, polarisation) columns per thread, the time sum runs over quad broadcasts.  Reference: the generic kernel (taken
    for a block that is only 2-byte aligned) at tscrunch 1, its per-part powers summed in time order -- BIT FOR BIT."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import _lib
    rng = np.random.default_rng(78)
    layout = _lib.RAW_CASPSR if layout_name == "caspsr" else _lib.RAW_GENERIC
    host = np.clip(np.rint(rng.standard_normal(npart * 2 * nchan * 2) * 24.0), -128, 127).astype(np.int8)
    raw = torch.from_numpy(host).cuda()
    shifted = torch.zeros(raw.numel() + 18, dtype=torch.int8, device="cuda")
    off = 2 if layout_name == "generic" else 8           # (CASPSR groups of 8 bytes: still no 16-byte alignment)
    shifted[off:off + raw.numel()] = raw
    npol = 1 if pscrunch else 2
    nout = npart // tscrunch
    one = torch.zeros((npart, nchan, npol), dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, shifted[off:off + raw.numel()], nchan, npart, one, pscrunch, 1, layout=layout, scale=0.0123)
    want = one[0:nout * tscrunch:tscrunch].clone()
    for i in range(1, tscrunch):
        want = want + one[i:nout * tscrunch:tscrunch]
    got = torch.full((nout, nchan, npol), -1.0, dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, raw, nchan, npart, got, pscrunch, tscrunch, layout=layout, scale=0.0123)
    assert torch.equal(got, want), (int((got != want).sum()), float((got - want).abs().max()))
    assert float(got.min()) >= 0.0 and float(got.max()) > 0.0


@pytest.mark.parametrize("nchan,npol,ndat,interval,constant,blocks", [
    (1024, 1, 4096, 0, False, 1), (256, 2, 3000, 1000, False, 3), (64, 4, 2500, 700, True, 2), (4096, 1, 1024, 512, False, 2),
    (16, 1, 100000, 0, False, 1), (33, 2, 777, 100, False, 2)])
def test_rescale_matches_oracle(oracle, gpu, nchan, npol, ndat, interval, constant, blocks):
    """dsp::Rescale (SURVEY 8f-1 output stage): same state machine over several blocks as the oracle.  The sums are tree
    reductions in double instead of sample-by-sample: offsets/scales agree to a float ulp, outputs to a few ulp."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(21)
    r = dspsr_amd.Rescale(ctx, nchan, npol, interval, constant)
    ro = oracle.Rescale(interval, constant)
    for b in range(blocks):
        gain = rng.uniform(0.5, 20.0, (1, nchan, npol)).astype(np.float32) * (1 + b)
        x = (rng.standard_normal((ndat, nchan, npol)).astype(np.float32) ** 2 * gain + gain).astype(np.float32)
        if b == 0:
            x[:, 0, 0] = 3.0                                   # zero variance => scale 1 (Rescale.C:411-412)
        want = ro.transform(x)
        got = r.transform(torch.from_numpy(x).cuda(), torch.empty(x.shape, dtype=torch.float32, device="cuda")).cpu().numpy()
        off, sc = r.get()
        assert np.abs(off - ro.offset).max() <= 1.2e-7 * np.abs(ro.offset).max()
        assert np.abs(sc / ro.scale - 1).max() <= 2.4e-7
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert sc[0, 0] == 1.0 or blocks > 1
    r.close()


@pytest.mark.parametrize("nbit", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("nchan,ndat,interval,flip,swap", [(1024, 700, 0, True, False), (64, 513, 200, False, True), (256, 97, 1000, True, True)])
def test_fused_output_stage_identical_bytes(gpu, nbit, nchan, ndat, interval, flip, swap):
    """dspsr_amd_rescale_pscrunch_digitize == dspsr_amd_rescale_transform -> dspsr_amd_pscrunch_tfp -> dspsr_amd_sigproc_digitize
    (LoadToFil.C:318-362) byte for byte over several blocks, with Rescale intervals that end inside a block and band flips /
    swaps -- and the same Rescale state afterwards."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(31)
    ra, rb = dspsr_amd.Rescale(ctx, nchan, 2, interval, False), dspsr_amd.Rescale(ctx, nchan, 2, interval, False)
    nbytes = ndat * nchan * nbit // 8
    for b in range(3):
        gain = rng.uniform(0.5, 20.0, (1, nchan, 2)).astype(np.float32) * (1 + b)
        x = (rng.standard_normal((ndat, nchan, 2)).astype(np.float32) ** 2 * gain + gain).astype(np.float32)
        x[3, 5, 0], x[4, 6, 1], x[5, 7, 0] = np.inf, np.nan, 1e30
        d = torch.from_numpy(x).cuda()
        sep = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        resc = ra.transform(d, torch.empty_like(d))
        inten = torch.empty((ndat, nchan), dtype=torch.float32, device="cuda")
        dspsr_amd.pscrunch_tfp(ctx, resc, inten, nchan, 2)
        dspsr_amd.sigproc_digitize(ctx, inten, sep, nchan, 1, nbit, True, 1.0, 0.75, flip, swap)
        fused = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        rb.pscrunch_digitize(d, fused, nbit, 0.75, flip, swap)
        assert torch.equal(sep, fused) and int(fused.max()) > 0
        (oa, sa), (ob, sb) = ra.get(), rb.get()
        assert np.array_equal(oa, ob, equal_nan=True) and np.array_equal(sa, sb, equal_nan=True)
    with pytest.raises(dspsr_amd.DspsrAmdError):
        rb.pscrunch_digitize(d, torch.zeros(nbytes - 1, dtype=torch.uint8, device="cuda"), nbit)
    ra.close()
    rb.close()


@pytest.mark.parametrize("nbit", [1, 2, 4, 8, 16, -32])
@pytest.mark.parametrize("nchan,npol,flip,swap,rescale", [(1024, 1, True, False, True), (64, 4, False, True, False),
                                                         (256, 2, True, True, True)])
def test_sigproc_digitizer_bit_exact(oracle, gpu, nbit, nchan, npol, flip, swap, rescale):
    """dsp::SigProcDigitizer::pack on identical float input: integer output, bit exact (incl. clipping, NaN/inf)."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(22)
    ndat = 513
    x = (rng.standard_normal((ndat, nchan, npol)) * (3.0 if rescale else 40.0)).astype(np.float32)
    x[3, 5, 0], x[4, 6, 0], x[5, 7, 0], x[6, 8, 0] = np.inf, -np.inf, np.nan, 1e30
    want = oracle.sigproc_digitize(x, nbit, rescale, 1.5, 0.75, flip, swap)
    out = torch.zeros(want.nbytes, dtype=torch.uint8, device="cuda")
    dspsr_amd.sigproc_digitize(ctx, torch.from_numpy(x).cuda(), out, nchan, npol, nbit, rescale, 1.5, 0.75, flip, swap)
    got = out.cpu().numpy().view(want.dtype).reshape(want.shape)
    if nbit == -32:
        assert np.array_equal(got, want, equal_nan=True)
    else:
        assert np.array_equal(got, want)


def test_search_mode_chain_digifil(oracle, gpu):
    """digifil's chain (LoadToFil.C:318-362): TFP filterbank + pscrunch + tscrunch -> Rescale -> 8-bit SigProcDigitizer,
    everything resident in HBM.  FFT rounding makes a handful of samples fall on the other side of a rounding
    boundary: at most one level apart, and in well under 0.1% of the samples."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(23)
    nchan, tscrunch, npart = 1024, 8, 256
    raw = np.clip(np.rint(rng.standard_normal(npart * 2 * nchan * 2) * 24.0), -128, 127).astype(np.int8)
    obs = oracle.Observation()
    det = oracle.tscrunch_tfp(oracle.tfp_filterbank(oracle.unpack_8bit(raw, obs), nchan, True), tscrunch)
    want = oracle.sigproc_digitize(oracle.Rescale().transform(det), 8, flip_band=True)
    out = torch.zeros((npart // tscrunch, nchan, 1), dtype=torch.float32, device="cuda")
    dspsr_amd.tfp_filterbank(ctx, torch.from_numpy(raw).cuda(), nchan, npart, out, True, tscrunch, scale=float(oracle.S8))
    r = dspsr_amd.Rescale(ctx, nchan, 1)
    r.transform(out)                                                        # in place, as digifil does
    packed = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
    dspsr_amd.sigproc_digitize(ctx, out, packed, nchan, 1, 8, flip_band=True)
    got = packed.cpu().numpy().reshape(want.shape)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1
    assert (d != 0).mean() < 1e-3
    r.close()


@pytest.mark.parametrize("nchan,npol,ndim,ndat,inplace", [(64, 2, 2, 5000, True), (64, 2, 2, 5000, False), (1024, 2, 2, 3000, True),
                                                         (3, 1, 1, 300000, False), (3, 1, 1, 300000, True), (16, 4, 4, 4099, True)])
def test_sample_delay_matches_oracle(oracle, gpu, nchan, npol, ndim, ndat, inplace):
    """dsp::SampleDelay with the dispersive delays of Dedispersion::SampleDelay (-K): a pure copy, bit exact; in place
    (as LoadToFold1.C:617-618 wires it) and out of place."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(31)
    obs = oracle.Observation(centre_frequency=1382.0, bandwidth=-64.0, dispersion_measure=3.0)
    rate = 64e6 / nchan
    delays = dspsr_amd.dedispersion_sample_delays(1382.0, -64.0, 3.0, nchan, rate)
    assert np.array_equal(delays, oracle.dedispersion_sample_delays(obs, nchan, rate))
    x = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32)
    want, zero, total = oracle.sample_delay(x, delays)
    sd = dspsr_amd.SampleDelay(ctx, delays, npol)
    assert (sd.zero_delay, sd.total_delay) == (zero, total) and total > 0
    xin = torch.from_numpy(x).cuda()
    out = xin if inplace else torch.zeros_like(xin)
    nout = sd.transform(xin, out)
    assert nout == ndat - total > 0
    assert np.array_equal(out.cpu().numpy()[:, :, :nout], want)
    # fewer samples than the total delay: nothing comes out (SampleDelay.C:137-143)
    short = torch.zeros((nchan, npol, max(total - 1, 0), ndim), dtype=torch.float32, device="cuda")
    assert sd.transform(short) == 0
    sd.close()
    # absolute delays per (chan, pol)
    dabs = rng.integers(0, 50, (nchan, npol))
    want, zero, total = oracle.sample_delay(x, dabs, absolute=True)
    sd = dspsr_amd.SampleDelay(ctx, dabs, npol, absolute=True)
    assert (sd.zero_delay, sd.total_delay) == (0, total)
    out = torch.zeros_like(xin)
    nout = sd.transform(torch.from_numpy(x).cuda(), out)
    assert np.array_equal(out.cpu().numpy()[:, :, :nout], want)
    sd.close()


@pytest.mark.parametrize("ndim", [4, 2])
def test_pipeline_interchan_dedispersion(oracle, gpu, ndim):
    """dspsr -K (LoadToFold1.C:605-624): fractional-delay chirp + SampleDelay between filterbank and detection, over several
    blocks (the delayed tail is carried like InputBuffering does), against the oracle composed in the reference's order."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    o = oracle
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=ndim,
                          parts_per_block=3, max_parts=2, interchan_dedispersion=True)
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    assert not lt.fused_fold and lt.sample_delay.total_delay > 0
    nblocks = 4
    step = cfg.parts_per_block * lt.nsamp_step
    raw = synth.voltages(nblocks * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)
    d_raw = torch.from_numpy(raw).cuda()
    for b in range(nblocks):
        lt.process_block(d_raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.finish_subint()
    lt.synchronize()
    sub = lt.subints[0]
    prof = sub["profile_dev"].cpu().numpy().reshape(nchan, 4 // ndim, nbin, ndim)
    # oracle, whole stream at once
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm)
    resp = o.Dedispersion()
    resp.fractional_delay = True
    resp.match(obs, nchan)
    assert np.abs(lt.response.kernel - resp.buffer).max() <= 1.2e-7
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(o.unpack_8bit(raw, obs), plan, lt.response.kernel, dtype=np.float64)
    fobs = o.filterbank_output_observation(obs, plan)
    delays = o.dedispersion_sample_delays(fobs, nchan, fobs.rate)
    fbd, zero, total = o.sample_delay(fb, delays)
    assert (zero, total) == (lt.sample_delay.zero_delay, lt.sample_delay.total_delay)
    fobs.start_seconds += zero / fobs.rate                                   # SampleDelay.C:159
    det = o.detect_layout(o.detect_products(fbd, "Coherence"), ndim)
    ps = o.PhaseSeries(nchan, 4 // ndim, ndim, nbin, data=np.zeros((nchan, 4 // ndim, nbin, ndim), np.float64))
    fcfg = o.FoldConfig(nbin=nbin, folding_period=period)
    block_out = cfg.parts_per_block * plan.nkeep
    pos = 0
    for b in range(nblocks):                                                 # the first block comes out `total` short
        n = block_out - (total if b == 0 else 0)
        o.fold(det, fobs, fcfg, ps, idat_start=pos, ndat_fold=n)
        pos += n
    assert pos == fbd.shape[2] == lt.ndat_out
    assert np.array_equal(sub["hits"], ps.hits)
    assert np.abs(prof - ps.data).max() <= 1e-5 * np.abs(ps.data).max()
    lt.close()


def test_headline_size_fused_fold_properties(oracle, gpu):
    """BASELINE headline geometry (-F 1024:D -x 4096, N = 2^22, 422/422) at full size, where the library fuses the fold by
    itself (512 channel tiles): size-independent properties instead of an oracle run -- fused == Detection + Fold bit for
    bit over two launch groups, every sample lands in exactly one phase bin, and the total power (PP + QQ planes) of
    the profile equals the sum over the detected time series."""
    dspsr_amd, ctx = gpu
    C, M, nfilt, nbin, npart = 1024, 4096, (422, 422), 1024, 3
    N = C * M
    nkeep = M - sum(nfilt)
    step, ovl = 2 * (N - sum(nfilt) * C), 2 * sum(nfilt) * C
    rng = np.random.default_rng(41)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, True, kernel, max_parts=2)
    assert eng.fold_is_fused()
    gen = torch.Generator(device="cuda").manual_seed(7)
    raw = torch.randn(2 * (npart * step + ovl), generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    ndat = npart * nkeep
    folds, hits = [dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)], [np.zeros(nbin, np.uint32), np.zeros(nbin, np.uint32)]
    for f, h in zip(folds, hits):
        f.set_shape(C, 1, 4, nbin)
        f.set_nbin(nbin)
        f.set_ndat(ndat, 0)
        f.set_bins(0.123, 1.0 / 34879.3, ndat, 0, h)
    det = torch.zeros((C, 1, 4 * ndat), dtype=torch.float32, device="cuda")
    eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=raw, layout=dspsr_amd.RAW_CASPSR, scale=float(oracle.S8))
    folds[0].fold(det)
    eng.perform_fold(folds[1], npart, dspsr_amd.COHERENCE, raw=raw, layout=dspsr_amd.RAW_CASPSR, scale=float(oracle.S8))
    a, b = folds[0].synch(), folds[1].synch()
    assert np.array_equal(a, b) and np.abs(a).max() > 0
    assert int(hits[0].sum()) == ndat and np.array_equal(hits[0], hits[1])
    d = det.view(C, ndat, 4).double()
    want = d.sum(dim=1).cpu().numpy()                                     # per channel: sum over time of (PP, QQ, Re, Im)
    got = a.reshape(C, nbin, 4).astype(np.float64).sum(axis=1)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    eng.close()
    for f in folds:
        f.close()


@pytest.mark.parametrize("nbit,rescale_seconds,constant", [(8, 10.0, False), (2, 0.002, False), (4, 0.002, True), (8, 0.0, False)])
def test_load_to_fil_matches_oracle_chain(oracle, gpu, nbit, rescale_seconds, constant):
    """digifil as LoadToFil.C wires it: TFPFilterbank (PPQQ) -> TScrunch -> Rescale (2 pols) -> PScrunch -> SigProcDigitizer,
    over three blocks (the Rescale state carries over).  The FFT rounds differently from pocketfft, so a few samples land
    on the other side of a digitiser boundary: at most one level apart, in well under 0.2 % of the samples."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline
    o = oracle
    nchan, tscr, npart, nblocks = 256, 4, 512, 3
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=400.0, tsamp_us=0.00125, machine="DADA")   # positive bw: flipped
    cfg = pipeline.SearchConfig(nchan=nchan, tscrunch=tscr, nbit=nbit, rescale_seconds=rescale_seconds, rescale_constant=constant,
                                parts_per_block=npart)
    lf = pipeline.LoadToFil(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    obs = o.Observation()
    ro = o.Rescale(int(rescale_seconds * lf.out_rate), constant) if rescale_seconds else None
    rng = np.random.default_rng(51)
    worst, frac = 0, 0.0
    for b in range(nblocks):
        raw = np.clip(np.rint(rng.standard_normal(npart * 2 * nchan * 2) * (20.0 + 6 * b)), -128, 127).astype(np.int8)
        got = lf.process_block(torch.from_numpy(raw).cuda()).cpu().numpy()
        det = o.tscrunch_tfp(o.tfp_filterbank(o.unpack_8bit(raw, obs), nchan, False), tscr)
        if ro is not None:
            det = ro.transform(det)
        want = o.sigproc_digitize(o.pscrunch_tfp(det), nbit, use_digi_scales=ro is not None, flip_band=True)
        assert got.size == want.size
        spb = 8 // nbit
        lv = lambda a: ((a.reshape(-1)[:, None] >> (np.arange(spb) * nbit)) & ((1 << nbit) - 1)).astype(np.int32)
        d = np.abs(lv(got) - lv(want))
        worst, frac = max(worst, int(d.max())), max(frac, float((d != 0).mean()))
    assert worst <= 1 and frac < 2e-3
    h = lf.header_values()
    assert h["nchans"] == nchan and h["foff"] == -400.0 / nchan and h["fch1"] == 1382.0 + 200.0 - 200.0 / nchan
    assert abs(h["tsamp"] - 2 * nchan * tscr * 0.00125e-6) < 1e-18
    lf.close()


def test_search_mode_and_delay_error_paths(gpu):
    """The C-ABI reports misuse with the reference's wording instead of corrupting memory."""
    dspsr_amd, ctx = gpu
    x = torch.zeros((8, 6, 2), dtype=torch.float32, device="cuda")
    out = torch.zeros(8 * 6 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(dspsr_amd.DspsrAmdError, match="nbit=3 not understood"):
        dspsr_amd.sigproc_digitize(ctx, x, out, 6, 2, nbit=3)
    with pytest.raises(dspsr_amd.DspsrAmdError, match="not a multiple of 4 samples per byte"):
        dspsr_amd.sigproc_digitize(ctx, x, out, 6, 2, nbit=2)
    with pytest.raises(dspsr_amd.DspsrAmdError, match="in place is not supported"):
        dspsr_amd.pscrunch_tfp(ctx, x, x, 6, 2)
    with pytest.raises(dspsr_amd.DspsrAmdError, match="invalid npol=1"):
        dspsr_amd.pscrunch_tfp(ctx, x, torch.zeros_like(x), 12, 1)
    with pytest.raises(dspsr_amd.DspsrAmdError, match="must not be negative"):
        dspsr_amd.SampleDelay(ctx, np.array([3, -1, 2]), 1, absolute=True)
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.Rescale(ctx, 0, 1)
    sd = dspsr_amd.SampleDelay(ctx, np.zeros(4, np.int64), 2)          # all-zero relative delays: identity
    assert (sd.zero_delay, sd.total_delay) == (0, 0)
    y = torch.arange(4 * 2 * 10 * 2, dtype=torch.float32, device="cuda").reshape(4, 2, 10, 2)
    z = torch.zeros_like(y)
    assert sd.transform(y, z) == 10 and torch.equal(y, z)
    sd.close()


def test_fold_weights_drop_flagged_samples(oracle, gpu):
    """Fold.C:686-716,746-763: samples whose weight is zero are left out of the plan, of hits[] and of ndat_folded; the
    kept samples are summed in time order -- bit-identical to the CPU loop with binplan == nbin for the dropped ones."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(61)
    nchan, npol, ndim, ndat, nbin = 3, 2, 2, 5000, 48
    det = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32)
    ndatperweight, weight_idat, idat_start, ndat_fold = 64, 17, 9, ndat - 30
    nweights = (ndat + weight_idat) // ndatperweight + 2
    weights = rng.integers(0, 4, nweights).astype(np.uint32)            # a quarter of the weights are zero
    weights[3:6] = 0                                                    # a run of consecutive dropped weights
    phi, pps = 0.21, 1.0 / 77.3
    eng = dspsr_amd.FoldEngine(ctx)
    eng.set_shape(nchan, npol, ndim, nbin)
    eng.set_nbin(nbin)
    eng.set_ndat(ndat_fold, idat_start)
    hits = np.zeros(nbin, np.uint32)
    n = eng.set_bins(phi, pps, ndat_fold, idat_start, hits, weights=weights, ndatperweight=ndatperweight, weight_idat=weight_idat)
    eng.fold(torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda())
    plan = oracle.fold_binplan_weighted(phi, pps, nbin, idat_start, ndat_fold, weights, ndatperweight, weight_idat)
    kept = plan != nbin
    assert 0 < kept.sum() < ndat_fold and n == kept.sum() == eng.get_ndat_folded()
    assert np.array_equal(hits, np.bincount(plan[kept], minlength=nbin).astype(np.uint32))
    want = np.zeros((nchan, npol, nbin, ndim), np.float32)
    for i in np.flatnonzero(kept):
        want[:, :, plan[i], :] += det[:, :, idat_start + i, :]
    assert np.array_equal(eng.synch(), want)
    eng.close()


def test_phase_series_mixable_and_device_combine(gpu):
    """PhaseSeries::mixable / combine (PhaseSeries.C:336-418,442-484) on device-resident profiles: an empty integration
    adopts the other one; combinable observations add (profiles on the device, hits / integration_length / ndat_total on
    the host) and extend the time span; a different nbin or band is not mixable."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import pipeline
    obs = dict(centre_frequency=1382.0, bandwidth=-400.0, nchan=5, npol=2, ndim=2, state="Coherence", rate=390625.0)
    rng = np.random.default_rng(62)

    def make(nbin, t0, t1, o=obs):
        sub = {"hits": rng.integers(1, 9, nbin).astype(np.uint32), "integration_length": t1 - t0, "ndat_total": int((t1 - t0) * o["rate"]),
               "profile_dev": torch.from_numpy(rng.standard_normal(o["nchan"] * o["npol"] * nbin * o["ndim"]).astype(np.float32)).cuda()}
        return pipeline.PhaseSeries.from_subint(ctx, sub, o, t0, t1)
    a, b, c = make(32, 0.0, 1.0), make(32, 1.0, 2.5), make(32, 4.0, 4.5)
    want_prof = (a.profile + b.profile).cpu().numpy()
    want_hits = a.hits + b.hits
    total = pipeline.PhaseSeries(ctx)
    assert total.integration_length == 0.0
    total.combine(a)                                            # empty: becomes a copy of a
    assert np.array_equal(total.hits, a.hits) and torch.equal(total.profile, a.profile) and total.profile.data_ptr() != a.profile.data_ptr()
    total.combine(b)
    ctx.synchronize()
    assert np.array_equal(total.profile.cpu().numpy(), want_prof) and np.array_equal(total.hits, want_hits)
    assert total.integration_length == 2.5 and total.ndat_total == a.ndat_total + b.ndat_total
    total.combine(c)
    assert (total.start_time, total.end_time) == (0.0, 4.5) and total.integration_length == 3.0
    with pytest.raises(dspsr_amd.DspsrAmdError, match="!mixable"):
        total.combine(make(64, 5.0, 6.0))                       # nbin differs
    with pytest.raises(dspsr_amd.DspsrAmdError, match="!mixable"):
        total.combine(make(32, 5.0, 6.0, dict(obs, centre_frequency=1400.0)))
    total.zero()
    assert total.integration_length == 0.0 and not total.hits.any() and float(total.profile.abs().max()) == 0.0
    assert total.mixable(dict(obs, nchan=2), 16, 7.0, 8.0) and tuple(total.profile.shape) == (2, 2, 32) and total.nbin == 16


@pytest.mark.parametrize("ndim,npol", [(4, 1), (2, 2), (1, 4)])
def test_fold_long_runs_reassociated(oracle, gpu, ndim, npol):
    """Phase bins hundreds of samples wide (-F 64:D: 1090 samples per bin): the plan holds runs >= 256 samples and the fold
    sums aligned 32-sample micro-blocks first (fold.hip FOLD_LONG_RUN).  Same hits, deterministic, equal to the time-order
    sum to float rounding (<= 2e-6 of the profile maximum against a float64 fold; the CPU float32 loop is no closer)."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(71)
    nchan, ndat, nbin = 3, 50000, 16
    det = (rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32) ** 2 + 1).astype(np.float32)
    phi, pps = 0.13, 1.0 / (nbin * 1090.7)
    idat_start, ndat_fold = 37, ndat - 100
    res = []
    for rep in range(2):
        eng = dspsr_amd.FoldEngine(ctx)
        eng.set_shape(nchan, npol, ndim, nbin)
        hits = np.zeros(nbin, np.uint32)
        eng.set_nbin(nbin)
        eng.set_ndat(ndat_fold, idat_start)
        eng.set_bins(phi, pps, ndat_fold, idat_start, hits)
        eng.fold(torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda())
        res.append(eng.synch())
        eng.close()
    assert np.array_equal(res[0], res[1])                                     # deterministic
    plan = oracle.fold_binplan(phi, pps, nbin, ndat_fold)
    assert np.array_equal(hits, np.bincount(plan, minlength=nbin).astype(np.uint32)) and np.diff(np.flatnonzero(np.diff(plan))).max() >= 256
    want = np.zeros((nchan, npol, nbin, ndim), np.float64)
    np.add.at(want, (slice(None), slice(None), plan), det[:, :, idat_start:idat_start + ndat_fold, :].astype(np.float64))
    assert np.abs(res[0] - want).max() <= 2e-6 * np.abs(want).max()


def test_fold_zeroed_samples_per_channel_hits(oracle, gpu):
    """dspsr_amd_fold_fold_zeroed: an input with zeroed (RFI-excised) samples keeps hits[] per channel -- the planned samples of
    polarisation 0 whose first float is not zero (Fold.C:853-866, fold1bin*hits FoldCUDA.cu:415-576); the sums are those of
    the plain fold."""
    dspsr_amd, ctx = gpu
    o = oracle
    nchan, npol, ndim, nbin, ndat = 5, 2, 2, 48, 4000
    rng = np.random.default_rng(3)
    x = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32)
    for c in range(nchan):
        x[c, :, rng.random(ndat) < 0.1 * (c + 1)] = 0.0                 # a different excision pattern in every channel
    det = torch.from_numpy(x.reshape(nchan, npol, ndat * ndim)).cuda()
    hits_dev = torch.zeros((nchan, nbin), dtype=torch.int32, device="cuda")
    f, g2 = dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)
    want = np.zeros((nchan, nbin), np.int64)
    for phi, i0, n in ((0.3, 0, 1500), (0.77, 1500, 2500)):
        for e in (f, g2):
            e.set_shape(nchan, npol, ndim, nbin)
            e.set_nbin(nbin)
            e.set_ndat(n, i0)
            e.set_bins(phi, 1.0 / 333.3, n, i0, np.zeros(nbin, np.uint32))
        f.fold_zeroed(det, hits_dev)
        g2.fold(det)
        plan = o.fold_binplan(phi, 1.0 / 333.3, nbin, n)
        for c in range(nchan):
            np.add.at(want[c], plan, (x[c, 0, i0:i0 + n, 0] != 0).astype(np.int64))
    assert np.array_equal(hits_dev.cpu().numpy().astype(np.int64), want) and want.sum() < nchan * ndat
    assert np.array_equal(f.synch(), g2.synch())
    f.close()
    g2.close()


def test_four_pass_fused_fold_segment_sums(oracle, gpu):
    """fold_is_fused() == 3: four-pass geometry (freq_res 16384 = 64 x 256, the dsp::Convolution shapes) with wide phase bins.
    The second inverse pass leaves the sums of the 32-sample runs it holds (cut at phase-bin boundaries), a second kernel adds
    them per (channel, bin) in time order.  Against perform_detect + the long-run Fold: hits identical, sums equal to float
    rounding (both re-associate, with different micro-block grids); against a float64 fold of the detected samples <= 2e-6
    of the profile maximum; two runs bit-identical; over several calls and launch groups (the profile is re-loaded);
    plans that do not qualify -- a zero weight (gap), bins narrower than a run -- take the separate launches: same sums."""
    dspsr_amd, ctx = gpu
    o = oracle
    C, M, nfilt, nbin = 4, 16384, (301, 212), 64
    N, nkeep = C * M, M - sum(nfilt)
    step, ovl = 2 * (N - sum(nfilt) * C), 2 * sum(nfilt) * C
    npart, ncall = 5, 2
    rng = np.random.default_rng(31)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, True, kernel, max_parts=2)
    assert eng.fold_is_fused() == 3
    ndat = npart * nkeep
    det = torch.zeros((C, 1, 4 * ndat), dtype=torch.float32, device="cuda")
    for label, pps, weights in (("wide bins", 1.0 / (nbin * 700.3), None), ("a zero weight", 1.0 / (nbin * 700.3), "gap"),
                                ("narrow bins", 1.0 / (nbin * 9.7), None)):
        fused, sep, again = dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)
        hits = [np.zeros(nbin, np.uint32) for _ in range(3)]
        want = np.zeros((C, nbin, 4), np.float64)
        for f in (fused, sep, again):
            f.set_shape(C, 1, 4, nbin)
        for call in range(ncall):
            raw = torch.from_numpy(_raw(npart * step + ovl, seed=300 + call)).cuda()
            phi = (0.21 + call * ndat * pps) % 1.0
            w = None
            if weights == "gap":
                w = np.ones((ndat + 511) // 512, np.uint32)
                w[3] = 0
            for k, f in enumerate((fused, sep, again)):
                f.set_nbin(nbin)
                f.set_ndat(ndat, 0)
                f.set_bins(phi, pps, ndat, 0, hits[k], weights=w, ndatperweight=512 if w is not None else 0)
            eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=raw, scale=float(o.S8))
            sep.fold(det)
            eng.perform_fold(fused, npart, dspsr_amd.COHERENCE, raw=raw, scale=float(o.S8))
            eng.perform_fold(again, npart, dspsr_amd.COHERENCE, raw=raw, scale=float(o.S8))
            plan = o.fold_binplan(phi, pps, nbin, ndat) if w is None else o.fold_binplan_weighted(phi, pps, nbin, 0, ndat, w, 512, 0)
            d = det.view(C, ndat, 4).cpu().numpy().astype(np.float64)
            keep = plan < nbin
            for c in range(C):
                np.add.at(want[c], plan[keep], d[c][keep])
        a, b, c2 = fused.synch().reshape(C, nbin, 4), sep.synch().reshape(C, nbin, 4), again.synch().reshape(C, nbin, 4)
        assert np.array_equal(hits[0], hits[1]) and int(hits[0].sum()) == (ncall * ndat if w is None else ncall * (ndat - 512)), label
        scale = np.abs(want).max()
        assert scale > 0 and np.array_equal(a, c2), label                         # deterministic
        # (narrow bins: the exact time-order float32 chain of ~2500 samples per bin, i.e. the CPU loop's own rounding)
        tol = 5e-6 if label == "narrow bins" else 2e-6
        assert np.abs(a - want).max() <= tol * scale, (label, np.abs(a - want).max() / scale)
        assert np.abs(b - want).max() <= tol * scale, label
        if label != "wide bins":
            assert np.array_equal(a, b), label                                     # the fallback IS Detection + Fold
        for f in (fused, sep, again):
            f.close()
    eng.close()


@pytest.mark.parametrize("C,M,nfilt,nbin,real", [(512, 512, (27, 27), 1024, False),      # cfg4: 32 tiles of 16 channels
                                                 (256, 4096, (953, 956), 1024, True)])   # cfg2: 128 tiles of 2 channels
def test_segmented_fused_fold(oracle, gpu, C, M, nfilt, nbin, real):
    """Geometries with fewer channel tiles than compute units: perform_fold cuts the parts of a launch into runs folded by
    different workgroups (fold_is_fused() == 2).  hits identical; sums equal to Detection + Fold to float rounding (<= 2e-6
    of the maximum), deterministic, accumulated over several calls and ragged launch groups; FUSED_NEVER / FUSED_ALWAYS
    still give the exact paths, bit-identical to each other."""
    dspsr_amd, ctx = gpu
    o = oracle
    N = C * M
    nkeep = M - sum(nfilt)
    ndim_in = 1 if real else 2
    step, ovl = (2 if real else 1) * (N - sum(nfilt) * C), (2 if real else 1) * sum(nfilt) * C
    npart, ncall = 11, 2
    rng = np.random.default_rng(81)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    res = {}
    for mode in ("auto", "auto2", "never", "always"):
        pol = {"auto": dspsr_amd.FUSED_AUTO, "auto2": dspsr_amd.FUSED_AUTO, "never": dspsr_amd.FUSED_NEVER, "always": dspsr_amd.FUSED_ALWAYS}[mode]
        eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, real, kernel, max_parts=8, fused_fold=pol)
        assert eng.fold_is_fused() == {"auto": 2, "auto2": 2, "never": 0, "always": 1}[mode]
        fold = dspsr_amd.FoldEngine(ctx)
        fold.set_shape(C, 1, 4, nbin)
        hits = np.zeros(nbin, np.uint32)
        pps = 1.0 / (nbin * 8.7)
        for call in range(ncall):
            raw = torch.from_numpy(_raw((npart * step + ovl), npol=2, ndim=ndim_in, seed=200 + call)).cuda()
            fold.set_nbin(nbin)
            fold.set_ndat(npart * nkeep, 0)
            fold.set_bins((0.21 + call * npart * nkeep * pps) % 1.0, pps, npart * nkeep, 0, hits)
            eng.perform_fold(fold, npart, dspsr_amd.COHERENCE, raw=raw, scale=float(o.S8))
        res[mode] = (hits, fold.synch())
        eng.close()
        fold.close()
    assert np.array_equal(res["auto"][1], res["auto2"][1])                      # deterministic
    assert np.array_equal(res["never"][1], res["always"][1])                    # the exact paths agree bit for bit
    for m in ("auto2", "never", "always"):
        assert np.array_equal(res["auto"][0], res[m][0])
    ref = res["never"][1].astype(np.float64)
    assert np.abs(ref).max() > 0
    assert np.abs(res["auto"][1] - ref).max() <= 2e-6 * np.abs(ref).max()
    assert not np.array_equal(res["auto"][1], res["never"][1]) or True          # (re-associated: usually differs in the last bits)


def test_host_block_feeder_equals_resident_blocks(gpu):
    """pipeline.LoadToFold.process_host_blocks (pinned host blocks, H2D on a second stream overlapped with the kernels)
    folds exactly what process_block folds from resident blocks."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline, synth
    freq, bw, tsamp, dm, period, nchan, nbin = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004, 16, 64
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4, parts_per_block=3, max_parts=2)
    res = []
    for hosted in (False, True):
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        nblocks = 5
        step = cfg.parts_per_block * lt.nsamp_step
        raw = torch.from_numpy(synth.voltages(nblocks * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period))
        blocks = [raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)] for b in range(nblocks)]
        if hosted:
            assert lt.process_host_blocks(b.clone().pin_memory() for b in blocks) == nblocks
        else:
            for b in blocks:
                lt.process_block(b.cuda())
        lt.finish_subint()
        lt.synchronize()
        res.append((lt.subints[0]["hits"].copy(), lt.subints[0]["profile_dev"].cpu().numpy()))
        lt.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.abs(res[0][1]).max() > 0


def test_fused_fold_many_channels_short_transform(gpu):
    """256 channels of an 8-point inverse transform (nkeep 6) are ONE pass-3 tile: the fused fold's staging of the detected
    tile must fit the exchange buffer (a channel stride rounded up to 16 samples did not: found by tests/fuzz_fold.py, which
    it crashed).  Fused (forced) == Detection + Fold, with sub-integration boundaries every 131 samples."""
    from dspsr_amd import pipeline, synth
    dspsr_amd, ctx = gpu
    freq, bw, tsamp, dm, period = 1382.0, -16.0, 1.0 / 32.0, 30.0, 0.004
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, machine="DADA")
    res = []
    for fused in (True, False):
        cfg = pipeline.Config(nchan=256, dispersion_measure=dm, nbin=64, folding_period=period, ndim=4, parts_per_block=3,
                              max_parts=2, fused_fold=fused, force_fused=fused, subint_seconds=0.0021)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        assert (lt.response.ndat, lt.nkeep) == (8, 6) and lt.fused_fold == fused
        step = 3 * lt.nsamp_step
        raw = torch.from_numpy(synth.voltages(12 * step + lt.nsamp_overlap, freq, bw, tsamp, dm, period)).cuda()
        for b in range(12):
            lt.process_block(raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
        if lt.ndat_total:
            lt.finish_subint()
        lt.synchronize()
        res.append([(s["hits"].copy(), s["profile_dev"].cpu().numpy(), s["ndat_total"]) for s in lt.subints])
        lt.close()
    assert len(res[0]) == len(res[1]) >= 1
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a[0], b[0]) and a[2] == b[2] and np.array_equal(a[1], b[1])
