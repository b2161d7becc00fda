"""The two-pass path of short responses (csrc/filterbank.hip, FB_HAS(6): k_raw_cols -> k_fwd_col1 -> k_rows_inv).

Complex dual-pol 8-bit input with nchan_subband * freq_res^2 == 2^27 (BASELINE cfg 4: one 50 MHz sub-band, -F 512:D -x 512):
the forward transform (2^14-point columns) and rows + chirp + inverse transforms in TWO workgroup tiles instead of three.
Reference arithmetic: Filterbank.C:561-662 (forward FFT, Response::operate, nchan_subband backward FFTs, keep window),
Detection (cross_detect.ic:23-43), Fold.C:835-891 -- the same oracle and tolerances as tests/test_gpu_parity.py:
  * filterbank output vs the float64 oracle: rms(err)/rms(out) <= 2e-6 * sqrt(log2 2N), max <= 8x that
  * two-pass vs three-pass (force_four_pass = 2) of the SAME library: both within that bound of the oracle, and within 4e-6 of
    each other relative to the rms (different association of the same transform)
  * fused fold with one workgroup per tile == perform_detect + fold, bit for bit (exact time order in both)
"""
import math

import numpy as np
import pytest

from test_gpu_parity import _fb_case, _raw, gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

# (nchan_subband, freq_res, (nfilt_pos, nfilt_neg)): Fb = 2^13 / M rows per inverse tile, Fa = C * M / Fb columns of pass 1.
# C * M^2 == 2^27 (Fa = 2^14): whole columns (k_raw_cols + k_fwd_col1); fewer channels (M <= Fa < 2^14): k_raw_transpose +
# k_fwd_cols on a geometry of their own, down to ONE tile of Fb channels (Fa = M)
GEOMETRIES = [(512, 512, (27, 27)), (128, 1024, (100, 61)), (32, 2048, (200, 301)), (8, 4096, (422, 400)),
              (256, 512, (27, 27)), (64, 512, (5, 90)), (16, 512, (40, 3)), (64, 1024, (100, 61)), (8, 2048, (2, 700)), (2, 4096, (422, 400))]


@pytest.mark.parametrize("C,M,nfilt", GEOMETRIES)
def test_two_pass_filterbank_against_oracle(oracle, gpu, C, M, nfilt):
    """Every geometry of the family, three parts in launch groups of two (a full and a ragged group), against the float64
    oracle; then the same input through the three-pass kernels of the same library."""
    got2, ref = _fb_case(oracle, gpu, C, M, nfilt, 3, npol=2, real=False, max_parts=2, seed=11)
    got3, _ = _fb_case(oracle, gpu, C, M, nfilt, 3, npol=2, real=False, max_parts=2, seed=11, four_pass=2)
    rms = math.sqrt(np.mean(np.abs(ref) ** 2))
    d = np.abs(got2 - got3)
    # (M = 4096: Fb = 2, and twiddle + radix-2 butterfly + chirp + the same inverse wgfft<12> are the very operations of
    #  k_fwd_rows<1> + k_inv_chan<12> in the same order: bit-identical by construction; that the path is taken is checked below)
    if M < 4096:
        assert d.max() > 0, "force_four_pass = 2 must take other kernels than the default (identical bits: the two-pass path did not run)"
    dspsr_amd, ctx = gpu
    e2 = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, False, None, max_parts=2)
    e3 = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, False, None, max_parts=2, force_four_pass=2)
    assert (e2.npass(True), e2.npass(False), e3.npass(True)) == (2, 3, 3)
    e2.close()
    e3.close()
    assert math.sqrt(np.mean(d ** 2)) <= 4e-6 * rms and d.max() <= 4e-5 * rms


def test_two_pass_multichannel_input(oracle, gpu):
    """Several input channels in the block (the NCHAN-8 full-band form of cfg 4, here 3 channels): the regroup pass picks one
    channel's words; every input channel against the oracle with its own slice of the kernel."""
    _fb_case(oracle, gpu, 512, 512, (27, 27), 2, npol=2, real=False, input_nchan=3, max_parts=2, seed=5)


def test_two_pass_float_input_takes_three_passes(oracle, gpu):
    """float32 rows (Filterbank::Engine::perform) are not regrouped per column: the object serves them with its three-pass
    kernels -- same object family, same oracle bound."""
    _fb_case(oracle, gpu, 128, 1024, (100, 61), 2, npol=2, real=False, max_parts=2, use_raw=False)


@pytest.mark.parametrize("ndim", [4, 2, 1])
@pytest.mark.parametrize("state", ["Coherence", "Stokes"])
def test_two_pass_detected_output(oracle, gpu, ndim, state):
    """perform_detect through the two-pass path: coherency / Stokes products (cross_detect.ic:23-43, stokes_detect.ic:21-44) of
    the filterbank output, against the float64 oracle's filterbank output detected in float64."""
    dspsr_amd, ctx = gpu
    o = oracle
    C, M, nfilt, npart = 128, 1024, (100, 61), 3
    N, nkeep = C * M, M - sum(nfilt)
    rng = np.random.default_rng(3)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    step, ovl = N - sum(nfilt) * C, sum(nfilt) * C
    raw = _raw(npart * step + ovl, 2, 2, 1, seed=8)
    obs = o.Observation(nchan=1, npol=2, ndim=2, machine="DADA")
    plan = o.FilterbankPlan(C, 1, C, M, N, nfilt[0], nfilt[1], sum(nfilt), N, ovl, step, nkeep, float(N) * M, False)
    ref = o.filterbank(o.unpack_8bit(raw, obs), plan, kernel, npart=npart, dtype=np.float64)       # [chan][pol][t] complex
    p, q = ref[:, 0], ref[:, 1]
    pp, qq, cr = np.abs(p) ** 2, np.abs(q) ** 2, p * np.conj(q)
    want = np.stack([pp + qq, pp - qq, 2 * cr.real, -2 * cr.imag] if state == "Stokes" else [pp, qq, cr.real, -cr.imag], axis=-1)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, False, kernel, max_parts=2)
    det = torch.zeros((C, 4 // ndim, ndim * npart * nkeep), dtype=torch.float32, device="cuda")
    st = dspsr_amd.STOKES if state == "Stokes" else dspsr_amd.COHERENCE
    eng.perform_detect(det, npart, st, ndim, raw=torch.from_numpy(raw).cuda(), scale=float(o.S8))
    eng.finish()
    got = det.cpu().numpy().reshape(C, 4 // ndim, npart * nkeep, ndim).transpose(0, 2, 1, 3).reshape(C, npart * nkeep, 4)
    eng.close()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("C,M,nfilt,nbin,period_samples", [(512, 512, (27, 27), 64, 71.3), (128, 1024, (100, 61), 128, 300.7),
                                                           (8, 4096, (422, 400), 32, 97.1), (64, 512, (27, 27), 64, 33.3),
                                                           (16, 1024, (100, 61), 16, 450.2)])
def test_two_pass_fused_fold_bit_identical(oracle, gpu, C, M, nfilt, nbin, period_samples):
    """perform_fold through k_rows_inv<., ., true> with one workgroup per tile for all parts of a launch (FUSED_ALWAYS: exact
    time order) == perform_detect + FoldEngine.fold (Fold.C:835-891), bit for bit, over several calls and launch groups."""
    dspsr_amd, ctx = gpu
    o = oracle
    N, nkeep = C * M, M - sum(nfilt)
    step, ovl = N - sum(nfilt) * C, sum(nfilt) * C
    npart, ncall = 5, 3
    rng = np.random.default_rng(23)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, False, kernel, max_parts=2, fused_fold=dspsr_amd.FUSED_ALWAYS)
    assert eng.fold_is_fused() == 1
    folds = [dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)]
    hits = [np.zeros(nbin, np.uint32), np.zeros(nbin, np.uint32)]
    for f in folds:
        f.set_shape(C, 1, 4, nbin)
    det = torch.zeros((C, 1, 4 * npart * nkeep), dtype=torch.float32, device="cuda")
    pps = 1.0 / period_samples
    for call in range(ncall):
        raw = torch.from_numpy(_raw(npart * step + ovl, 2, 2, 1, seed=100 + call)).cuda()
        ndat = npart * nkeep
        phi = (0.37 + call * ndat * pps) % 1.0
        for k, f in enumerate(folds):
            f.set_nbin(nbin)
            f.set_ndat(ndat, 0)
            f.set_bins(phi, pps, ndat, 0, hits[k])
        eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=raw, scale=float(o.S8))
        folds[0].fold(det)
        eng.perform_fold(folds[1], npart, dspsr_amd.COHERENCE, raw=raw, scale=float(o.S8))
    a, b = folds[0].synch(), folds[1].synch()
    assert np.array_equal(hits[0], hits[1]) and int(hits[0].sum()) == ncall * npart * nkeep
    assert np.abs(a).max() > 0 and np.array_equal(a, b)
    eng.close()
    for f in folds:
        f.close()


def test_two_pass_segmented_fused_fold_and_pipeline(oracle, gpu):
    """The cfg 4 shard as the bench runs it (pipeline.LoadToFold on sub-band g of an NCHAN-8 band, segmented fused fold: the
    32 tiles do not fill the chip) against (a) the same pipeline with Detection and Fold as separate launches (<= 2e-6 of the
    profile maximum: the part runs of the segmented fold are re-associated) and (b) the three-pass pipeline (two_pass=False),
    identical hits in every case."""
    dspsr_amd, _ = gpu
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=8, npol=2, ndim=2, tsamp_us=0.02, machine="DADA")
    res = {}
    for key, kw in (("two_fused", {}), ("two_unfused", {"fused_fold": False}), ("three_fused", {"two_pass": False})):
        cfg = pipeline.Config(nchan=4096, dispersion_measure=1000.0, nbin=1024, folding_period=0.0893, freq_res=512, ndim=4,
                              parts_per_block=24, max_parts=24, **kw)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, subband=5)
        raw = torch.from_numpy(_raw(lt.block_bytes() // 4, 2, 2, 1, seed=77)).cuda()
        for _ in range(2):
            lt.process_block(raw)
        lt.finish_subint()
        lt.synchronize()
        s = lt.subints[0]
        res[key] = (s["hits"].copy(), pipeline.subint_profile(s).reshape(-1), lt.fused_fold)
        lt.close()
    assert res["two_fused"][2] and not res["two_unfused"][2]
    h, p, _ = res["two_fused"]
    assert int(h.sum()) == 2 * 24 * 458 and np.abs(p).max() > 0
    for key in ("two_unfused", "three_fused"):
        assert np.array_equal(res[key][0], h), key
        assert np.abs(res[key][1] - p).max() <= 2e-6 * np.abs(p).max(), (key, np.abs(res[key][1] - p).max() / np.abs(p).max())


# ------------------------------------------------------------------------------------------------------------------------------
# nchan_subband = 3 * 2^k / 5 * 2^k (dspsr -F 96:D, -F 320:D; Filterbank.C:107-155 plans any length): the forward transform as 3 / 5
# interleaved power-of-two sub-sequences + one radix-3 / radix-5 step (k_sub_split, k_sub_combine), the inverse pass on
# nsub << logR rows.  Same oracle (numpy's FFT takes any length), same bounds.
@pytest.mark.parametrize("C,M,nfilt,kw", [
    (96, 256, (20, 21), dict()),                                   # real dual-pol 8-bit, -F 96:D
    (96, 256, (20, 21), dict(use_raw=False)),                      # float32 rows
    (80, 512, (40, 30), dict(real=False)),                         # complex dual-pol, 5 * 16 channels
    (48, 1024, (100, 101), dict(layout="caspsr")),                 # CASPSR byte order
    (12, 4096, (422, 400), dict(max_parts=2)),                     # few channels, long response
    (192, 64, (5, 7), dict(real=False, use_raw=False)),            # complex float rows
    (24, 128, (9, 10), dict(real=False, input_nchan=2)),           # two input channels
    (320, 128, (9, 10), dict(npol=1)),                             # single polarisation, 5 * 64
    # any odd factor up to 127 (round 5) (round 5: run-time-radix combine, k_sub_combine_any): dspsr -F 400:D, -F 25:D -x 2048, primes too
    (400, 256, (20, 21), dict()),                                  # 25 * 16
    (25, 2048, (300, 211), dict(max_parts=3)),                     # one channel per sub-sequence
    (44, 512, (40, 30), dict(real=False)),                         # 11 * 4, complex dual-pol
    (52, 128, (9, 10), dict(layout="caspsr")),                     # 13 * 4
    (63, 256, (20, 21), dict(use_raw=False)),                      # 63 sub-sequences of one channel, float32 rows
    (42, 1024, (100, 101), dict(real=False, input_nchan=2)),       # 21 * 2, two input channels
    (1000, 64, (5, 7), dict()),                                    # -F 1000:D: 125 * 8
    (254, 256, (20, 21), dict(real=False)),                        # the largest factor, a prime: 127 * 2
])
def test_filterbank_three_and_five_times_power_of_two_channels(oracle, gpu, C, M, nfilt, kw):
    _fb_case(oracle, gpu, C, M, nfilt, 3, max_parts=kw.pop("max_parts", 2), seed=9, **kw)


@pytest.mark.parametrize("C,M,nfilt,kw", [
    (3 * 4096, 4096, (422, 400), dict()),                          # L = 3 * 2^25, 8192 rows per sub-spectrum: the largest three-pass rows
    (25 * 256, 4096, (422, 400), dict(layout="caspsr")),           # run-time-radix combine at L = 25 * 2^21
    (16, 45 * 2048, (9000, 8000), dict()),                         # freq_res = 45 * 2^11 = 92160: long pseudo-channel responses
])
def test_odd_factor_twiddles_at_large_lengths(oracle, gpu, C, M, nfilt, kw):
    """the radix-R twiddles come from v_sin / v_cos on a split argument (fb_common.h twiddle_odd): the float64 oracle at
    lengths near the largest the three-pass form accepts, same tolerance as every other geometry (2e-6 sqrt(log2 2N) rms)."""
    _fb_case(oracle, gpu, C, M, nfilt, 1, seed=10, **kw)


def test_non_power_of_two_channels_fused_fold_bit_identical(oracle, gpu):
    """perform_fold == perform_detect + fold, bit for bit, with 96 channels (fused kernel, one workgroup per tile)."""
    dspsr_amd, ctx = gpu
    o = oracle
    C, M, nfilt, nbin, npart = 96, 512, (40, 30), 64, 4
    N, nkeep = C * M, M - sum(nfilt)
    step, ovl = 2 * (N - sum(nfilt) * C), 2 * sum(nfilt) * C
    rng = np.random.default_rng(5)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt[0], nfilt[1], 1, 2, True, kernel, max_parts=3, fused_fold=dspsr_amd.FUSED_ALWAYS)
    assert (eng.nsamp_fft, eng.nkeep) == (2 * N, nkeep)
    folds = [dspsr_amd.FoldEngine(ctx), dspsr_amd.FoldEngine(ctx)]
    hits = [np.zeros(nbin, np.uint32), np.zeros(nbin, np.uint32)]
    for f in folds:
        f.set_shape(C, 1, 4, nbin)
    det = torch.zeros((C, 1, 4 * npart * nkeep), dtype=torch.float32, device="cuda")
    raw = torch.from_numpy(_raw(npart * step + ovl, seed=3)).cuda()
    for k, f in enumerate(folds):
        f.set_nbin(nbin)
        f.set_ndat(npart * nkeep, 0)
        f.set_bins(0.21, 1.0 / 45.3, npart * nkeep, 0, hits[k])
    eng.perform_detect(det, npart, dspsr_amd.COHERENCE, 4, raw=raw, scale=float(o.S8))
    folds[0].fold(det)
    eng.perform_fold(folds[1], npart, dspsr_amd.COHERENCE, raw=raw, scale=float(o.S8))
    a, b = folds[0].synch(), folds[1].synch()
    assert np.abs(a).max() > 0 and np.array_equal(a, b) and np.array_equal(hits[0], hits[1])
    eng.close()
    for f in folds:
        f.close()


def test_rejected_channel_counts(gpu):
    dspsr_amd, ctx = gpu
    for C in (129, 131, 2 * 129, 255, 4 * 243):              # odd factor above 127
        with pytest.raises(dspsr_amd.DspsrAmdError):
            dspsr_amd.FilterbankEngine(ctx).setup(C, 256, 5, 5, 1, 2, True, None)


# ---------------------------------------------------------------------------------------------------------------------------
# freq_res = 3 * 2^k / 5 * 2^k (dspsr -x 12288; the reference plans any length, Filterbank.C:107-155): R pseudo-channels of
# freq_res / R bins per channel through the power-of-two passes, then one radix-R step in time (k_time_combine)
@pytest.mark.parametrize("C,M,nfilt,npart,kw", [
    (16, 3 * 256, (50, 41), 3, dict(max_parts=2)),                      # real dual-pol 8-bit
    (8, 5 * 128, (33, 20), 2, dict()),                                  # radix 5
    (64, 3 * 512, (100, 101), 2, dict(real=False)),                     # complex dual-pol: no mirror rows
    (32, 3 * 64, (7, 9), 2, dict(npol=1)),                              # one polarisation
    (4, 3 * 4096, (1000, 900), 2, dict(layout="caspsr")),               # -x 12288, CASPSR byte order
    (16, 5 * 256, (60, 60), 3, dict(use_raw=False, max_parts=2)),       # float32 rows, ragged launch groups
    (8, 3 * 128, (20, 21), 2, dict(input_nchan=2, real=False)),         # two input channels
    (128, 3 * 2, (1, 1), 2, dict()),                                    # the shortest inner transform
    (1, 3 * 4096, (500, 400), 2, dict()),                               # one channel (dsp::Convolution shape)
    (2, 5 * 8192, (3000, 2000), 1, dict(real=False)),                   # the longest inner transform
    (16, 7 * 64, (30, 21), 2, dict()),                                  # other small odd factors: 7, 9, 15
    (8, 9 * 128, (100, 50), 2, dict(real=False)),
    (32, 15 * 32, (40, 41), 3, dict(max_parts=2)),
    (56, 256, (20, 21), 2, dict()),                                     # ... of the channel count (-F 56:D, -F 36:D, -F 60:D)
    (36, 512, (40, 30), 2, dict(real=False)),
    (60, 128, (9, 10), 3, dict(layout="caspsr", max_parts=2)),
    (96, 3 * 256, (50, 41), 2, dict()),                                 # both lengths at once: -F 96:D -x 768 (9 sub-sequences)
    (80, 3 * 128, (20, 21), 2, dict(real=False)),                       # 5 x 3
    # part steps that are not a multiple of the factor: the launch group runs as 3 / 5 interleaved sub-groups of several parts
    # (round 4: one part per launch), every part a window of its own in the de-interleaved block
    (16, 3 * 256, (50, 41), 9, dict(max_parts=7)),
    (16, 5 * 256, (60, 61), 11, dict(use_raw=False, max_parts=8)),
    (8, 3 * 512, (60, 61), 7, dict(layout="caspsr", max_parts=7)),
    (32, 9 * 64, (30, 21), 8, dict(real=False, max_parts=8)),
    # any odd factor up to 127 (round 5) of freq_res (k_sub_combine_any<true>, k_time_combine<0>): dspsr -x 11264, -x 6400
    (16, 11 * 1024, (700, 600), 2, dict()),                             # -x 11264
    (8, 25 * 256, (301, 300), 5, dict(max_parts=5)),                    # -x 6400, part step not a multiple of 25: five sub-groups
    (32, 13 * 64, (30, 21), 3, dict(real=False)),
    (4, 49 * 32, (100, 50), 2, dict(layout="caspsr")),
    (16, 63 * 16, (20, 21), 2, dict(use_raw=False)),
    (24, 15 * 64, (40, 41), 2, dict()),                                 # 3 (channels) x 15 = 45
    (80, 5 * 64, (9, 10), 2, dict(real=False)),                         # 5 x 5 = 25
    (8, 125 * 64, (500, 411), 3, dict(max_parts=3)),                    # -x 8000
    (24, 7 * 64 * 5, (40, 41), 2, dict()),                              # 3 x 35 = 105
])
def test_filterbank_freq_res_three_five_times_power_of_two(oracle, gpu, C, M, nfilt, npart, kw):
    _fb_case(oracle, gpu, C, M, nfilt, npart, **kw)


def test_freq_res_not_power_of_two_detected_and_folded(oracle, gpu):
    """The same geometry through perform_detect (every ndim / state) and perform_fold (separate Detection + Fold launches:
    fold_is_fused() == 0) against the oracle's filterbank + detection + fold."""
    dspsr_amd, ctx = gpu
    o = oracle
    C, M, nfilt_pos, nfilt_neg, npart = 16, 3 * 256, 50, 41, 3
    N = C * M
    rng = np.random.default_rng(5)
    kernel = np.exp(1j * rng.uniform(-np.pi, np.pi, N)).astype(np.complex64)
    obs = o.Observation(nchan=1, npol=2, ndim=1, machine="DADA")
    plan = o.FilterbankPlan(C, 1, C, M, N, nfilt_pos, nfilt_neg, nfilt_pos + nfilt_neg, 2 * N, 2 * (nfilt_pos + nfilt_neg) * C, 0,
                            M - nfilt_pos - nfilt_neg, float(N) * M, True)
    plan.nsamp_step = plan.nsamp_fft - plan.nsamp_overlap
    ndat = npart * plan.nsamp_step + plan.nsamp_overlap
    raw = np.clip(np.rint(rng.standard_normal(ndat * 2) * 24.0), -128, 127).astype(np.int8)
    ref = o.filterbank(o.unpack_8bit(raw, obs), plan, kernel, npart=npart, dtype=np.float64)       # [chan][pol][t] complex
    eng = dspsr_amd.FilterbankEngine(ctx).setup(C, M, nfilt_pos, nfilt_neg, 1, 2, True, kernel, max_parts=2)
    assert eng.fold_is_fused() == 0 and eng.nkeep == plan.nkeep
    d_raw = torch.from_numpy(raw).cuda()
    nout = npart * plan.nkeep
    p, q = ref[:, 0, :], ref[:, 1, :]
    coh = np.stack([np.abs(p) ** 2, np.abs(q) ** 2, (p * np.conj(q)).real, -(p * np.conj(q)).imag], axis=-1)     # cross_detect.ic:23-43
    scale = np.abs(coh).max()
    for ndim in (4, 2, 1):
        det = torch.zeros((C, 4 // ndim, nout * ndim), dtype=torch.float32, device="cuda")
        eng.perform_detect(det, npart, dspsr_amd.COHERENCE, ndim, raw=d_raw, scale=float(o.S8))
        eng.finish()
        g = det.cpu().numpy().reshape(C, 4 // ndim, nout, ndim).transpose(0, 2, 1, 3).reshape(C, nout, 4)
        assert np.abs(g - coh).max() <= 2e-5 * scale, (ndim, np.abs(g - coh).max() / scale)
    # fold: 64 bins, against the time-order sums of the oracle's detected samples
    nbin = 64
    fold = dspsr_amd.FoldEngine(ctx)
    fold.set_shape(C, 1, 4, nbin)
    fold.set_nbin(nbin)
    fold.set_ndat(nout, 0)
    hits = np.zeros(nbin, np.uint32)
    fold.set_bins(0.3, 1.0 / 97.3, nout, 0, hits)
    eng.perform_fold(fold, npart, dspsr_amd.COHERENCE, raw=d_raw, scale=float(o.S8))
    binplan = o.fold_binplan(0.3, 1.0 / 97.3, nbin, nout)
    want = np.zeros((C, nbin, 4))
    for i in range(nout):
        want[:, binplan[i], :] += coh[:, i, :]
    got = fold.synch().reshape(C, nbin, 4)
    assert np.array_equal(hits, np.bincount(binplan, minlength=nbin).astype(np.uint32))
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    fold.close()
    eng.close()


def test_freq_res_other_lengths_are_refused(gpu):
    dspsr_amd, ctx = gpu
    for C, M in ((16, 129 * 64), (16, 255 * 64), (24, 7 * 64 * 3 * 5 * 3), (80, 27 * 64), (16, 3)):    # odd factors above 127 (together)
        with pytest.raises(dspsr_amd.DspsrAmdError):
            dspsr_amd.FilterbankEngine(ctx).setup(C, M, 1, 1, 1, 2, True, None)


def test_freq_res_not_power_of_two_through_the_pipeline(gpu):
    """LoadToFold with -F 16:D -x 3072 (Detection and Fold as separate launches): every output sample lands in a phase bin and the
    folded profile equals the sum of the detected block the pipeline keeps, bin by bin in time order."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-64.0, nchan=1, npol=2, ndim=1, tsamp_us=1.0 / 128.0, machine="DADA")
    cfg = pipeline.Config(nchan=16, dispersion_measure=10.0, nbin=64, folding_period=0.00123, freq_res=3072, ndim=4, parts_per_block=6, max_parts=4)
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    assert not lt.fused_fold and lt.fb.npass(True) == 3 and lt.nkeep == 3072 - lt.response.impulse_pos - lt.response.impulse_neg
    raw = torch.from_numpy(_raw(lt.block_bytes() // 2, 2, 1, 1, seed=21)).cuda()
    lt.process_block(raw)
    lt.finish_subint()
    lt.synchronize()
    sub = lt.subints[-1]
    ndat = 6 * lt.nkeep
    assert int(sub["hits"].sum()) == ndat
    prof = sub["profile_dev"].cpu().numpy().reshape(16, 64, 4)
    det = lt.detected.view(16, ndat, 4).cpu().numpy()
    assert np.isfinite(prof).all() and np.abs(prof).max() > 0
    assert np.allclose(prof.sum(axis=1), det.sum(axis=1), rtol=2e-4, atol=1e-3 * np.abs(det).max())
    lt.close()
