"""GPU parity at the sizes and the launch shape bench.py times.

bench.py runs the headline geometry (-F 1024:D -x 4096, DM 1000, N = 2^22) with 64 parts per block and 32 parts per
launch group; inside a group pass 2 and the inverse pass are interleaved in 8-part sub-groups.  These tests run exactly
that shape through the same driver (pipeline.LoadToFold) and check it
  (a) bit for bit against the one-part-per-launch path (no grouping, no sub-groups), fused and unfused,
  (b) against the float64 oracle on a handful of channels over ALL parts: detected samples and folded profile <= 1e-5
      of the profile maximum, hits identical, folding with a constant period and with the vela.polyco predictor.
The remaining BASELINE configurations that had no GPU case get one each: cfg3 (DM 2000, 843/844) and cfg1 at its optimal
response length (64 x 262144, N = 2^24).
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHANS = [0, 1, 510, 511, 1023]          # first tile (both channels), a middle tile, the last channel


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import dspsr_amd
    return dspsr_amd


def _headline(dspsr_amd, max_parts, fused, polyco=None, parts_per_block=64, stokes=False):
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125,
                              machine="CASPSR")
    cfg = pipeline.Config(nchan=1024, dispersion_measure=1000.0, nbin=1024, folding_period=0.0 if polyco else 0.0893,
                          freq_res=4096, ndim=4, parts_per_block=parts_per_block, max_parts=max_parts, fused_fold=fused,
                          stokes=stokes)
    return pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, polyco=polyco)


def _noise_block(nbytes, seed=20100413):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    raw = torch.empty(nbytes, dtype=torch.int8, device="cuda")
    chunk = 1 << 26
    for s in range(0, nbytes, chunk):
        e = min(nbytes, s + chunk)
        raw[s:e] = torch.randn(e - s, generator=gen, device="cuda").mul_(24.0).round_().clamp_(-128, 127).to(torch.int8)
    return raw


def _oracle_detected(o, raw_host, plan, kernel, chans, npart, batch=4, state="Coherence"):
    """float64 oracle of the detected (Coherence, ndim 4) samples of `chans` for all parts:
    Filterbank.C:561-662 + Response.C:385-444 + cross_detect.ic restated -- only the selected channels are
    inverse-transformed (the forward transforms are the cost: 2 x npart real FFTs of 2^23 points, batched over cores)."""
    import scipy.fft
    obs = o.Observation(machine="CASPSR")
    N, M, nk, nfp = plan.n_fft, plan.freq_res, plan.nkeep, plan.nfilt_pos
    out = np.zeros((len(chans), 1, npart * nk, 4), np.float64)
    k64 = kernel.astype(np.complex128)
    workers = max(1, len(os.sched_getaffinity(0)))
    for p0 in range(0, npart, batch):
        nb = min(batch, npart - p0)
        x = np.empty((nb, 2, plan.nsamp_fft), np.float64)
        for j in range(nb):
            lo = 2 * (p0 + j) * plan.nsamp_step                     # 2 bytes per sample (two polarisations)
            x[j] = o.unpack_8bit(raw_host[lo: lo + 2 * plan.nsamp_fft], obs)[0].astype(np.float64)
        spec = scipy.fft.rfft(x, axis=2, workers=workers)[:, :, :N]          # frc1d: first N of the N+1 bins
        for ci, c in enumerate(chans):
            t = np.fft.ifft(spec[:, :, c * M:(c + 1) * M] * k64[c * M:(c + 1) * M], axis=2) * M   # unnormalised bcc1d
            t = t[:, :, nfp:nfp + nk]
            for j in range(nb):
                prod = o.detect_products(t[j][None, :, :], state)            # [1][4][nkeep], float64 in -> float64
                out[ci, 0, (p0 + j) * nk:(p0 + j + 1) * nk, :] = o.detect_layout(prod, 4)[0, 0]
    return out


def test_headline_timed_launch_shape(oracle, gpu):
    dspsr_amd = gpu
    o = oracle
    from dspsr_amd import pipeline
    polyco_text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
    npart = 64
    lt = _headline(dspsr_amd, 32, True)
    assert lt.fused_fold and (lt.response.impulse_pos, lt.response.impulse_neg, lt.nkeep) == (422, 422, 3252)
    raw = _noise_block(lt.block_bytes())
    ndat = npart * lt.nkeep

    def run(ltx):
        ltx.process_block(raw)
        ltx.finish_subint()
        ltx.synchronize()
        sub = ltx.subints[-1]
        return sub["hits"].copy(), sub["profile_dev"].cpu().numpy().reshape(1024, 1024, 4), sub["integration_length"]

    # (a) the timed shape == one part per launch, bit for bit; fused == Detection + Fold as separate operations
    h32, p32, t32 = run(lt)
    lt1 = _headline(dspsr_amd, 1, True)
    h1, p1, _ = run(lt1)
    lt1.close()
    assert np.array_equal(h32, h1) and np.array_equal(p32, p1) and np.abs(p32).max() > 0
    ltu = _headline(dspsr_amd, 32, False)
    assert not ltu.fused_fold
    hu, pu, _ = run(ltu)
    assert np.array_equal(h32, hu) and np.array_equal(p32, pu)
    det_dev = ltu.detected.view(1024, ndat, 4)[CHANS].cpu().numpy().astype(np.float64)      # [chan][ndat][4]
    ltu1 = _headline(dspsr_amd, 1, False)
    ltu1.process_block(raw)
    ltu1.synchronize()
    assert torch.equal(ltu1.detected, ltu.detected)                  # the whole detected block, every channel and part
    ltu1.close()
    ltu.close()
    assert int(h32.sum()) == ndat and t32 == ndat / lt.out_rate

    # (b) float64 oracle on a handful of channels over all parts
    obs = o.Observation(centre_frequency=1382.0, bandwidth=-400.0, tsamp_us=0.00125, machine="CASPSR", dispersion_measure=1000.0)
    resp = o.Dedispersion()
    resp.set_frequency_resolution(4096)
    resp.match(obs, 1024)
    plan = o.filterbank_plan(obs, 1024, resp)
    assert (plan.nkeep, plan.nsamp_step) == (lt.nkeep, lt.nsamp_step)
    assert np.abs(lt.response.kernel - resp.buffer).max() <= 1.2e-7
    det = _oracle_detected(o, raw.cpu().numpy(), plan, lt.response.kernel, CHANS, npart)
    dmax = np.abs(det[..., :2]).max()
    assert np.abs(det_dev - det[:, 0]).max() <= 1e-5 * dmax, np.abs(det_dev - det[:, 0]).max() / dmax
    fobs = o.filterbank_output_observation(obs, plan)
    assert fobs.rate == lt.out_rate and fobs.start_seconds == lt.out_start
    # constant period
    ps = o.PhaseSeries(len(CHANS), 1, 4, 1024, data=np.zeros((len(CHANS), 1, 1024, 4), np.float64))
    o.fold(det, fobs, o.FoldConfig(nbin=1024, folding_period=0.0893), ps)
    assert np.array_equal(h32, ps.hits)
    err = np.abs(p32[CHANS] - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max()
    assert err <= 1e-5, err
    lt.close()
    # vela.polyco predictor (phase and period from the polynomial at the first sample, Fold.C:943-958)
    ltp = _headline(dspsr_amd, 32, True, polyco=pipeline.Polyco(polyco_text))
    hp, pp, _ = run(ltp)
    ltp.close()
    ps = o.PhaseSeries(len(CHANS), 1, 4, 1024, data=np.zeros((len(CHANS), 1, 1024, 4), np.float64))
    o.fold(det, fobs, o.FoldConfig(nbin=1024, polyco=o.Polyco.parse(polyco_text)), ps)
    assert np.array_equal(hp, ps.hits) and not np.array_equal(hp, h32)
    err = np.abs(pp[CHANS] - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max()
    assert err <= 1e-5, err


def test_headline_ragged_groups_and_stokes(gpu):
    """Group sizes that do not divide: 45 parts in launch groups of 32 (13 left over) and sub-groups of 8 (5 left over),
    Stokes products; fused == unfused == one part per launch, bit for bit."""
    dspsr_amd = gpu
    res = []
    raw = None
    for max_parts, fused in ((32, True), (1, True), (32, False)):
        lt = _headline(dspsr_amd, max_parts, fused, parts_per_block=45, stokes=True)
        if raw is None:
            raw = _noise_block(lt.block_bytes(), seed=5)
        lt.process_block(raw)
        lt.process_block(raw)                     # accumulators are re-loaded from the device profile
        lt.finish_subint()
        lt.synchronize()
        res.append((lt.subints[0]["hits"].copy(), lt.subints[0]["profile_dev"].cpu().numpy()))
        lt.close()
    for h, p in res[1:]:
        assert np.array_equal(h, res[0][0]) and np.array_equal(p, res[0][1])
    assert int(res[0][0].sum()) == 2 * 45 * 3252 and np.abs(res[0][1]).max() > 0


def test_headline_stokes_against_oracle(oracle, gpu):
    """dspsr -4 (Stokes I, Q, U, V: stokes_detect.ic:21-44) at the headline geometry through the fused launch group, against
    the float64 oracle on five channels: hits identical, folded Stokes profile <= 1e-5 of the profile maximum."""
    o = oracle
    npart = 8
    lt = _headline(gpu, 8, True, parts_per_block=npart, stokes=True)
    assert lt.fused_fold
    raw = _noise_block(lt.block_bytes(), seed=11)
    lt.process_block(raw)
    lt.finish_subint()
    lt.synchronize()
    sub = lt.subints[0]
    prof = sub["profile_dev"].cpu().numpy().reshape(1024, 1024, 4)
    obs = o.Observation(centre_frequency=1382.0, bandwidth=-400.0, tsamp_us=0.00125, machine="CASPSR", dispersion_measure=1000.0)
    resp = o.Dedispersion()
    resp.set_frequency_resolution(4096)
    resp.match(obs, 1024)
    plan = o.filterbank_plan(obs, 1024, resp)
    det = _oracle_detected(o, raw.cpu().numpy(), plan, lt.response.kernel, CHANS, npart, state="Stokes")
    fobs = o.filterbank_output_observation(obs, plan)
    ps = o.PhaseSeries(len(CHANS), 1, 4, 1024, data=np.zeros((len(CHANS), 1, 1024, 4), np.float64))
    o.fold(det, fobs, o.FoldConfig(nbin=1024, folding_period=0.0893), ps)
    assert np.array_equal(sub["hits"], ps.hits) and int(ps.hits.sum()) == npart * lt.nkeep
    # Stokes I is positive; Q, U, V are differences / cross terms of noise: every product against the I scale
    err = np.abs(prof[CHANS] - ps.data[:, 0]).max() / np.abs(ps.data[..., 0]).max()
    assert err <= 1e-5, err
    assert np.abs(prof[CHANS][..., 1:]).max() > 0
    lt.close()


def test_cfg1_pipeline_against_oracle(oracle, gpu):
    """BASELINE configuration 1 as the pipeline really runs it: dspsr -F 64:D -x 16384 on the header.dada band, vela.par DM,
    512 phase bins, vela.polyco -- the FOUR-pass path (two-pass inverse, Convolution.C:338-461 geometry), Detection, and
    the long-run fold (1090 samples per phase bin: re-associated micro-block sums, Fold.C:835-891 to rounding); then the same
    with the segment-sum fold forced (fold_is_fused() == 3, what cfg1 at its optimal response length takes).
    Against the float64 oracle on five channels: hits identical, folded profile <= 1e-5 of the profile maximum."""
    o = oracle
    from dspsr_amd import pipeline
    polyco_text = json.load(open(os.path.join(ROOT, "tests", "golden", "vela_polyco.json")))["text"]
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125, machine="CASPSR")
    cfg = pipeline.Config(nchan=64, dispersion_measure=67.99, nbin=512, folding_period=0.0, freq_res=16384, ndim=4,
                          parts_per_block=6, max_parts=4)
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream, polyco=pipeline.Polyco(polyco_text))
    assert lt.fused_mode == 0     # four-pass geometry; only 1817 of 16384 samples are kept: Detection + the long-run Fold
    nblocks, chans = 2, [0, 1, 31, 62, 63]
    step = cfg.parts_per_block * lt.nsamp_step
    raw = _noise_block(2 * (nblocks * step + lt.nsamp_overlap), seed=23)
    for b in range(nblocks):
        lt.process_block(raw[2 * b * step: 2 * (b * step + step + lt.nsamp_overlap)])
    lt.finish_subint()
    lt.synchronize()
    sub = lt.subints[0]
    prof = sub["profile_dev"].cpu().numpy().reshape(64, 512, 4)
    obs = o.Observation(centre_frequency=1382.0, bandwidth=-400.0, tsamp_us=0.00125, machine="CASPSR", dispersion_measure=67.99)
    resp = o.Dedispersion()
    resp.set_frequency_resolution(16384)
    resp.match(obs, 64)
    plan = o.filterbank_plan(obs, 64, resp)
    assert (plan.nkeep, plan.nsamp_step) == (lt.nkeep, lt.nsamp_step)
    npart = nblocks * cfg.parts_per_block
    det = _oracle_detected(o, raw.cpu().numpy(), plan, lt.response.kernel, chans, npart)
    fobs = o.filterbank_output_observation(obs, plan)
    ps = o.PhaseSeries(len(chans), 1, 4, 512, data=np.zeros((len(chans), 1, 512, 4), np.float64))
    # the pipeline folds block by block (phase and period from the polynomial at each block's first sample, Fold.C:943-958)
    nblk = cfg.parts_per_block * lt.nkeep
    for b in range(nblocks):
        o.fold(det, fobs, o.FoldConfig(nbin=512, polyco=o.Polyco.parse(polyco_text)), ps, idat_start=b * nblk, ndat_fold=nblk)
    assert np.array_equal(sub["hits"], ps.hits) and int(ps.hits.sum()) == npart * lt.nkeep
    assert ps.hits.max() > 64                                  # wide phase bins (1090 samples each): the long-run fold is what ran
    err = np.abs(prof[chans] - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max()
    assert err <= 1e-5, err
    lt.close()
    ltf = pipeline.LoadToFold(pipeline.Config(**{**cfg.__dict__, "force_fused": True}), info, device=0,
                              stream=torch.cuda.current_stream().cuda_stream, polyco=pipeline.Polyco(polyco_text))
    assert ltf.fused_mode == 3
    for b in range(nblocks):
        ltf.process_block(raw[2 * b * step: 2 * (b * step + step + ltf.nsamp_overlap)])
    ltf.finish_subint()
    ltf.synchronize()
    subf = ltf.subints[0]
    assert np.array_equal(subf["hits"], ps.hits)
    err = np.abs(subf["profile_dev"].cpu().numpy().reshape(64, 512, 4)[chans] - ps.data[:, 0]).max() / np.abs(ps.data[..., :2]).max()
    assert err <= 1e-5, err
    ltf.close()


def _fb(oracle, gpu_mod, *a, **k):
    from test_gpu_parity import _fb_case
    ctx = gpu_mod.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        return _fb_case(oracle, (gpu_mod, ctx), *a, **k)
    finally:
        ctx.close()


def test_filterbank_cfg3_size(oracle, gpu):
    # BASELINE cfg3: -F 1024:D -x 4096, DM 2000 on the header.dada band -> 843/844, nkeep 2409
    _fb(oracle, gpu, 1024, 4096, (843, 844), 2, layout="caspsr", max_parts=2)


def test_filterbank_cfg1_optimal_size(oracle, gpu):
    # BASELINE cfg1 at the optimal response length: -F 64:D, freq_res 262144 (N = 2^24), 7226/7341, nkeep 247577
    _fb(oracle, gpu, 64, 262144, (7226, 7341), 1, layout="caspsr")


def test_filterbank_cfg1_optimal_size_paired_pass1_tiles(oracle, gpu):
    # the same geometry over three parts in launch groups of two and one: L = 2^25 -> pass 1 on PAIRS of two-column tiles
    # (k_fwd_cols_dual: the outputs of the first tile wait in registers, four columns are copied out together in whole cache
    # lines); every (pair, part) item against the float64 oracle
    _fb(oracle, gpu, 64, 262144, (7226, 7341), 3, layout="caspsr", max_parts=2, seed=11)


def test_paired_pass1_tiles_complex_input(oracle, gpu):
    # complex dual-polarisation input of N = 2^25 points per polarisation: two sequences per part (nseq = 2) through the paired
    # pass-1 tiles (k_fwd_cols_dual walks (pair, sequence, part) items), two channels of 2^24 bins each
    _fb(oracle, gpu, 2, 1 << 24, (600000, 500000), 1, real=False, seed=12)


def test_cfg3_pipeline_geometry(gpu):
    """cfg3 through the driver: the host preparation yields the Appendix-B numbers and the fused fold conserves power."""
    dspsr_amd = gpu
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(machine="CASPSR")
    cfg = pipeline.Config(nchan=1024, dispersion_measure=2000.0, nbin=1024, folding_period=0.0893, freq_res=4096,
                          parts_per_block=16, max_parts=8)
    lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    assert (lt.response.impulse_pos, lt.response.impulse_neg, lt.nkeep, lt.nsamp_step) == (843, 844, 2409, 4933632)
    raw = _noise_block(lt.block_bytes(), seed=3)
    lt.process_block(raw)
    lt.finish_subint()
    ltu = pipeline.LoadToFold(pipeline.Config(**{**cfg.__dict__, "fused_fold": False}), info, device=0,
                              stream=torch.cuda.current_stream().cuda_stream)
    ltu.process_block(raw)
    ltu.finish_subint()
    a, b = lt.subints[0], ltu.subints[0]
    assert np.array_equal(a["hits"], b["hits"]) and torch.equal(a["profile_dev"], b["profile_dev"])
    want = ltu.detected.view(1024, -1, 4).double().sum(dim=1)
    got = a["profile_dev"].view(1024, 1024, 4).double().sum(dim=1)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    lt.close()
    ltu.close()


def test_filterbank_very_long_response(oracle, gpu):
    """Towards the upper end of the supported range: single-channel coherent dedispersion (dsp::Convolution geometry) with a
    2^24-point response on real dual-pol input -- forward transform of 2^25 points (8192 x 4096), two-pass inverse
    4096 x 4096."""
    _fb(oracle, gpu, 1, 1 << 24, (700000, 650000), 1, layout="caspsr")


def test_fold_many_bins_direct_kernel(oracle, gpu):
    """nbin beyond what the chunked kernel's workgroup covers (4096 bins): the direct fold kernel, still bit-identical to
    the CPU loop."""
    ctx = gpu.Context(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(91)
    nchan, npol, ndim, ndat, nbin = 2, 1, 4, 30000, 6000
    det = rng.standard_normal((nchan, npol, ndat, ndim)).astype(np.float32)
    eng = gpu.FoldEngine(ctx)
    eng.set_shape(nchan, npol, ndim, nbin)
    eng.set_nbin(nbin)
    eng.set_ndat(ndat, 0)
    hits = np.zeros(nbin, np.uint32)
    eng.set_bins(0.4, 1.0 / 7001.3, ndat, 0, hits)
    eng.fold(torch.from_numpy(det.reshape(nchan, npol, ndat * ndim)).cuda())
    plan = oracle.fold_binplan(0.4, 1.0 / 7001.3, nbin, ndat)
    want = np.zeros((nchan, npol, nbin, ndim), np.float32)
    for i in range(ndat):
        want[:, :, plan[i], :] += det[:, :, i, :]
    assert np.array_equal(hits, np.bincount(plan, minlength=nbin).astype(np.uint32)) and np.array_equal(eng.synch(), want)
    eng.close()
    ctx.close()


@pytest.mark.parametrize("C,M,nfilt,npart,kw", [
    (1024, 4096, (422, 422), 2, dict(four_pass=True, max_parts=2)),          # the headline band through the four-pass path
    (8, 1 << 19, (30000, 29000), 1, dict()),                                # freq_res 2^19: 2048 x 256 inverse split
    (4, 1 << 20, (50000, 60000), 1, dict(real=False)),                      # complex dual-pol input: two sequences of L = N
    (64, 1 << 15, (2000, 2100), 2, dict(npol=1)),                           # single polarisation
    (32, 1 << 16, (4000, 4100), 3, dict(use_raw=False, max_parts=2)),       # float input, ragged launch groups
    (32, 1 << 16, (4000, 4100), 1, dict(input_nchan=2)),                    # two input channels: kernel slice per channel
    (4096, 512, (40, 41), 1, dict(four_pass=True)),                         # many channels, short inverse (32 x 16)
    (1, 1 << 22, (100000, 90000), 1, dict(layout="caspsr")),                # one channel (dsp::Convolution), L = 2^23
])
def test_four_pass_blocked_spectrum_geometries(oracle, gpu, C, M, nfilt, npart, kw):
    """L >= 2^22 in four-pass mode: the spectrum between pass 2 and the inverse is blocked by pass-2 tile, k_inv_a loads it in
    memory order (thread / item / per-bit address parts, mirror bin from the complement), the chirp is permuted likewise on
    upload.  Every variant of input the path takes, against the float64 oracle."""
    _fb(oracle, gpu, C, M, nfilt, npart, **kw)


@pytest.mark.parametrize("C,M,nfilt,npart,kw", [
    (64, 1024, (100, 101), 3, dict(max_parts=2)),                      # real dual-pol, three passes, 16-column tiles
    (1024, 4096, (422, 422), 2, dict(max_parts=2)),                    # the headline geometry, 4-column tiles
    (128, 512, (40, 30), 2, dict(real=False)),                         # complex dual-pol: one sequence per polarisation
    (32, 2048, (100, 50), 2, dict(input_nchan=2)),                     # two input channels (channel stride)
])
def test_float_input_regrouped(oracle, gpu, C, M, nfilt, npart, kw):
    """dsp::Filterbank::Engine::perform receives float32 rows (DSPSR unpacks before the boundary).  With rows of >= 64
    columns the floats are first regrouped per pass-1 tile (k_float_transpose, 8-byte elements) exactly like the 8-bit
    stream; same results as the 8-bit side channel to rounding, and within tolerance of the float64 oracle."""
    a, _ = _fb(oracle, gpu, C, M, nfilt, npart, use_raw=False, **kw)
    b, _ = _fb(oracle, gpu, C, M, nfilt, npart, use_raw=True, **kw)
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()


def test_fused_fold_of_wide_bins(oracle, gpu):
    """Phase bins 64 ... 640 samples wide are still folded inside the last filterbank pass (exact time order: one dependent
    chain per bin), the separate Fold takes them as long runs (re-associated micro-block sums): the two agree to rounding,
    hits are identical, and the fused sums equal the CPU loop's order on the detected samples bit for bit."""
    from dspsr_amd import pipeline
    info = pipeline.InputInfo(machine="CASPSR")
    res = {}
    for fused in (True, False):
        cfg = pipeline.Config(nchan=1024, dispersion_measure=1000.0, nbin=128, folding_period=0.0893, freq_res=4096,
                              parts_per_block=8, max_parts=4, fused_fold=fused)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        assert lt.fused_fold == fused and 34883 // 128 > 64
        raw = _noise_block(lt.block_bytes(), seed=11)
        lt.process_block(raw)
        lt.finish_subint()
        lt.synchronize()
        sub = lt.subints[0]
        res[fused] = (sub["hits"].copy(), sub["profile_dev"].cpu().numpy().reshape(1024, 128, 4))
        if not fused:
            det = lt.detected.view(1024, -1, 4)[:3].cpu().numpy()                 # three channels, all samples
            t0 = lt.out_start + 0.5 / lt.out_rate
            plan = oracle.fold_binplan(math.fmod(t0, 0.0893) / 0.0893, (1.0 / lt.out_rate) / 0.0893, 128, det.shape[1])
        lt.close()
    assert np.array_equal(res[True][0], res[False][0]) and int(res[True][0].sum()) == 8 * 3252
    scale = np.abs(res[False][1]).max()
    assert np.abs(res[True][1] - res[False][1]).max() <= 2e-6 * scale
    want = np.zeros((3, 128, 4), np.float32)
    for i in range(det.shape[1]):                                                  # Fold.C:844-852: strict time order
        want[:, plan[i], :] += det[:, i, :]
    assert np.array_equal(res[True][1][:3], want)
