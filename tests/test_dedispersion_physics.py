"""A physical anchor for the conventions the oracle restates (FFT sign, chirp sign, band sense, channel order and
channel centre frequencies are "parity unpinned" against reference-built code): a pulse-modulated noise signal is dispersed with the
cold-plasma transfer function of the WHOLE band -- H(f) = exp(-i 2 pi D f^2 / (fc^2 (fc + f))), D = DM / 2.41e-4 MHz^2 s, whose group
delay is the textbook t(nu) - t(fc) = D (1 / nu^2 - 1 / fc^2) (dspsr_amd/synth.py; written from the dispersion law, not from
Dedispersion.C) -- and pushed through the restated chain.  If every convention is right,

  * `-F N:D` (Config::During) and `-F N` (Config::After: non-convolving filterbank, then dsp::Convolution matched to its output)
    give a pulse as sharp as the intrinsic one in EVERY channel,
  * at the phase the dispersion law predicts for the channel's centre frequency -- (c + 1/2) channel widths from the band edge
    behind the convolving filterbank, c widths behind the non-convolving one (its channels are centred on the transform's bins:
    Filterbank.C:341-348 sets dc_centred for freq_res = 1), in natural FFT order (band swapped) for complex-sampled input
    (Filterbank.C:358-364), running down in frequency for a negative bandwidth,
  * `-F N:B` (Config::Before: the whole band dedispersed first, then channelised) gives sharp pulses at ONE phase in all channels,
  * and without the response (Config::Never) the pulse stays smeared.

A wrong sign, a missing or doubled band swap, a half-channel error in the chirp's reference frequency or an inverted band each
break one of the three.  CPU: the float64 oracle; GPU (marked): the product's LoadToFold in the same three modes."""
import math

import numpy as np
import pytest

DM_CONST = 2.41e-4


def _expected_bins(freq, bw, nchan, dm, period, nbin, complex_input, dc_centred):
    chbw = bw / nchan
    out = []
    for c in range(nchan):
        cc = (c + nchan // 2) % nchan if complex_input else c            # natural FFT order of a dual-sideband band
        nu = freq - 0.5 * bw + (cc + (0.0 if dc_centred else 0.5)) * chbw
        delay = dm / DM_CONST * (1.0 / nu ** 2 - 1.0 / freq ** 2)         # seconds, relative to the band centre
        out.append(((0.5 * period + delay) / period) % 1.0 * nbin)       # the envelope peaks at phase 0.5 (synth._pulse_envelope)
    return np.array(out)


def _profile(o, fb, rate, start_seconds, period, nbin):
    """Total intensity PP + QQ folded at the constant period, per channel, mean removed."""
    prod = o.detect_products(fb, "Coherence")
    inten = prod[:, 0] + prod[:, 1]
    nd = inten.shape[1]
    plan = o.fold_binplan(((start_seconds + 0.5 / rate) / period) % 1.0, (1.0 / rate) / period, nbin, nd)
    hits = np.bincount(plan, minlength=nbin)
    prof = np.stack([np.bincount(plan, weights=inten[c], minlength=nbin) / hits for c in range(inten.shape[0])])
    return prof - prof.mean(axis=1, keepdims=True)


def _check(prof, expected, nbin, skip, sharp):
    for c in range(prof.shape[0]):
        if c in skip:
            continue
        d = (prof[c].argmax() + 0.5 - expected[c] + nbin / 2) % nbin - nbin / 2
        assert abs(d) <= 1.0, (c, prof[c].argmax(), expected[c])
        s = prof[c].max() / prof[c].std()
        assert (s >= 4.0) if sharp else (s <= 3.2), (c, s)


CASES = [("real lower sideband", 1, -8.0), ("complex lower sideband", 2, -8.0), ("complex upper sideband", 2, 8.0)]


@pytest.mark.parametrize("name,ndim,bw", CASES)
def test_dispersion_law_pins_the_conventions_oracle(oracle, name, ndim, bw):
    o = oracle
    from dspsr_amd import synth
    freq, dm, period, nchan, nbin = 1382.0, 150.0, 0.004, 8, 64
    tsamp = 1.0 / 16.0 if ndim == 1 else 1.0 / 8.0
    ndat = 1 << 19 if ndim == 1 else 1 << 18                            # 32.8 ms: eight pulses
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period, ndim=ndim, pulse_amp=8.0)
    obs = o.Observation(centre_frequency=freq, bandwidth=bw, tsamp_us=tsamp, dispersion_measure=dm, ndim=ndim)
    un = o.unpack_8bit(raw, obs)
    cplx = ndim == 2
    # intra-channel smearing at this DM: 0.47 ms of a 4 ms period = 7.5 of 64 bins; the intrinsic pulse is 3.2 bins wide
    # -F N:D
    resp = o.Dedispersion().match(obs, nchan)
    plan = o.filterbank_plan(obs, nchan, resp)
    fb = o.filterbank(un, plan, resp.buffer, dtype=np.float64)
    fobs = o.filterbank_output_observation(obs, plan)
    assert fobs.swap == cplx and not fobs.dc_centred
    during = _profile(o, fb, fobs.rate, fobs.start_seconds, period, nbin)
    _check(during, _expected_bins(freq, bw, nchan, dm, period, nbin, cplx, False), nbin, skip=(), sharp=True)
    # -F N: filterbank, then convolution matched to the filterbank's output
    plan1 = o.filterbank_plan(obs, nchan, None, 1)
    fb1 = o.filterbank(un, plan1, None, dtype=np.float64)
    obs1 = o.filterbank_output_observation(obs, plan1)
    assert obs1.dc_centred and obs1.get_dual_sideband() and obs1.swap == cplx
    r1 = o.Dedispersion().match(obs1)
    cv = o.convolution(np.ascontiguousarray(fb1).view(np.float64), r1.ndat, r1.impulse_pos, r1.impulse_neg, r1.buffer, False,
                       dtype=np.float64)
    after = _profile(o, cv, obs1.rate, obs1.start_seconds + r1.impulse_pos / obs1.rate, period, nbin)
    exp1 = _expected_bins(freq, bw, nchan, dm, period, nbin, cplx, True)
    # (real input: channel 0 of the non-convolving filterbank is the band edge at DC, half a channel of one-sided spectrum;
    #  complex input: the channel nchan/2 that straddles the two band edges)
    edge = (nchan // 2,) if cplx else (0,)
    _check(after, exp1, nbin, skip=edge, sharp=True)
    # no response: the same channels, smeared
    never = _profile(o, fb1, obs1.rate, obs1.start_seconds, period, nbin)
    sharp = lambda q: np.array([q[c].max() / q[c].std() for c in range(nchan) if c not in edge])
    assert (sharp(never) < 0.8 * sharp(after)).all(), (sharp(never), sharp(after))
    # -F N:B (Config::Before): the whole band dedispersed to its centre frequency first, then channelised: every channel's pulse
    # sharp AND at the phase of the band centre -- no delay between the channels is left
    rb = o.Dedispersion()
    rb.set_frequency_resolution(65536)                                   # (the optimal length, 2^19, is longer than the test signal)
    rb.match(obs)
    cb = o.convolution(un, rb.ndat, rb.impulse_pos, rb.impulse_neg, rb.buffer, ndim == 1, dtype=np.float64)
    obs_c = o.Observation(**obs.__dict__)
    obs_c.ndim, obs_c.tsamp_us = 2, obs.tsamp_us * (2 if ndim == 1 else 1)
    planb = o.filterbank_plan(obs_c, nchan, None, 1)
    fbb = o.filterbank(np.ascontiguousarray(cb).view(np.float64), planb, None, dtype=np.float64)
    obsb = o.filterbank_output_observation(obs_c, planb)
    before = _profile(o, fbb, obsb.rate, obs.start_seconds + rb.impulse_pos / obs_c.rate, period, nbin)
    _check(before, np.full(nchan, 0.5 * nbin), nbin, skip=edge, sharp=True)
    # a response with the wrong ordering (the per-channel swap of Response::match left out) does not sharpen
    bad = r1.buffer.reshape(nchan, 2, -1)[:, ::-1, :].reshape(-1).copy()
    cvb = o.convolution(np.ascontiguousarray(fb1).view(np.float64), r1.ndat, r1.impulse_pos, r1.impulse_neg, bad, False, dtype=np.float64)
    wrong = _profile(o, cvb, obs1.rate, obs1.start_seconds + r1.impulse_pos / obs1.rate, period, nbin)
    assert np.median(sharp(wrong)) < 0.85 * np.median(sharp(after)), (sharp(wrong), sharp(after))


@pytest.mark.gpu
@pytest.mark.parametrize("name,ndim,bw", CASES)
def test_dispersion_law_pins_the_conventions_product(name, ndim, bw):
    """The same through the HIP path: pipeline.LoadToFold with convolve_when = during / after / never on the 8-bit block."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from dspsr_amd import pipeline, synth
    freq, dm, period, nchan, nbin = 1382.0, 150.0, 0.004, 8, 64
    tsamp = 1.0 / 16.0 if ndim == 1 else 1.0 / 8.0
    ndat = 1 << 19 if ndim == 1 else 1 << 18
    raw = synth.voltages(ndat, freq, bw, tsamp, dm, period, ndim=ndim, pulse_amp=8.0)
    d_raw = torch.from_numpy(raw).cuda()
    info = pipeline.InputInfo(centre_frequency=freq, bandwidth=bw, ndim=ndim, tsamp_us=tsamp, machine="DADA")
    cplx = ndim == 2
    edge = (nchan // 2,) if cplx else (0,)
    bps = 2 * ndim                                                       # bytes per time sample (two polarisations)
    sharpness = {}
    for when in ("during", "after", "before", "never"):
        cfg = pipeline.Config(nchan=nchan, dispersion_measure=dm, nbin=nbin, folding_period=period, ndim=4,
                              parts_per_block=1 if when == "before" else 4, max_parts=2, convolve_when=when,
                              freq_res=65536 if when == "before" else 0)
        lt = pipeline.LoadToFold(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
        step = cfg.parts_per_block * lt.nsamp_step
        nblocks = (ndat - lt.nsamp_overlap) // step
        assert nblocks >= 2
        for b in range(nblocks):
            lt.process_block(d_raw[bps * b * step: bps * (b * step + step + lt.nsamp_overlap)])
        lt.finish_subint()
        lt.synchronize()
        s = lt.subints[0]
        p = s["profile_dev"].cpu().numpy().reshape(nchan, nbin, 4).astype(np.float64)
        prof = (p[:, :, 0] + p[:, :, 1]) / s["hits"][None, :]
        prof -= prof.mean(axis=1, keepdims=True)
        exp = _expected_bins(freq, bw, nchan, dm, period, nbin, cplx, when != "during") if when != "before" else np.full(nchan, 0.5 * nbin)
        sharpness[when] = np.array([prof[c].max() / prof[c].std() for c in range(nchan) if c not in edge])
        if when != "never":
            _check(prof, exp, nbin, skip=() if when == "during" else edge, sharp=True)
        lt.close()
    assert (sharpness["never"] < 0.8 * sharpness["after"]).all(), sharpness
