"""digifil with the convolving filterbank, `digifil -F N:D -t T -b nbit` (the branch of LoadToFil.C:185-222 next to the
TFP filterbank): Filterbank -> Detection::square_law -> [FScrunch] -> TScrunch -> Rescale -> SigProcDigitizer in FPT order.
The HIP path (through the C-ABI) against the oracle's restatements (oracle/dspsr_oracle.py: square_law, tscrunch_fpt, fscrunch_fpt,
Rescale, sigproc_digitize_fpt, DigifilCoherent), bit for bit wherever the arithmetic is the same."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dspsr_amd
    ctx = dspsr_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    yield dspsr_amd, ctx
    ctx.close()


@pytest.fixture(scope="module")
def oracle():
    import oracle.dspsr_oracle as o
    return o


@pytest.mark.parametrize("nchan,npol,sf,blocks", [(5, 1, 16, (1000, 37, 3, 2048)), (3, 2, 7, (100, 5, 1, 1, 9, 640)), (64, 1, 1, (33, 10)),
                                                  (2, 2, 300, (299, 1, 1000, 7))])
def test_tscrunch_fpt_stream_bit_exact(oracle, gpu, nchan, npol, sf, blocks):
    """dsp::TScrunch::fpt_tscrunch over a stream of blocks whose lengths are not multiples of the factor: the partial sum of the
    left-over samples is carried (the reference re-presents the samples, TScrunch.C:110-111) -- the same sequential sums, bit for bit."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(41)
    x = (rng.standard_normal((nchan, npol, sum(blocks))).astype(np.float32) ** 2) * 100
    want = oracle.tscrunch_fpt(x, sf)
    carry = torch.zeros((nchan, npol), dtype=torch.float32, device="cuda")
    cc, got, pos = 0, [], 0
    for n in blocks:
        blk = torch.from_numpy(np.ascontiguousarray(x[:, :, pos:pos + n])).cuda()
        out = torch.full((nchan, npol, (cc + n) // sf + 1), -1.0, dtype=torch.float32, device="cuda")
        nout, cc = dspsr_amd.tscrunch_fpt(ctx, blk, out, sf, carry, cc)
        assert nout == (pos + n) // sf - pos // sf and cc == (pos + n) % sf
        got.append(out[:, :, :nout].cpu().numpy())
        pos += n
    got = np.concatenate(got, axis=2)
    assert got.shape == want.shape and np.array_equal(got, want)
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.tscrunch_fpt(ctx, blk, blk, sf, carry, 0)                 # in place: refused


def test_tscrunch_fpt_carry_read_before_it_is_replaced(oracle, gpu):
    """A block that begins AND ends inside an output sample, with more outputs per row than one workgroup holds and thousands of rows:
    output 0 reads the carry, the open group at the end replaces it.  (Round 5: the two were different threads -- on the GPU the new
    carry could land before output 0 had read the old one; tests/fuzz_search.py 300 702 case 246, 1729 rows wrong.  One thread does both.)"""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(43)
    nchan, npol, sf = 2048, 2, 3
    blocks = (274, 822, 1000, 2)
    x = (rng.standard_normal((nchan, npol, sum(blocks))).astype(np.float32) ** 2) * 100
    want = oracle.tscrunch_fpt(x, sf)
    for rep in range(3):
        carry = torch.zeros((nchan, npol), dtype=torch.float32, device="cuda")
        cc, got, pos = 0, [], 0
        for n in blocks:
            blk = torch.from_numpy(np.ascontiguousarray(x[:, :, pos:pos + n])).cuda()
            out = torch.full((nchan, npol, (cc + n) // sf + 1), -1.0, dtype=torch.float32, device="cuda")
            nout, cc = dspsr_amd.tscrunch_fpt(ctx, blk, out, sf, carry, cc)
            got.append(out[:, :, :nout].cpu().numpy())
            pos += n
        got = np.concatenate(got, axis=2)
        assert got.shape == want.shape and np.array_equal(got, want)


def test_fscrunch_fpt_bit_exact(oracle, gpu):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(42)
    x = rng.standard_normal((24, 2, 1001)).astype(np.float32) ** 2
    for sf in (2, 3, 8):
        out = torch.zeros((24 // sf, 2, 1001), dtype=torch.float32, device="cuda")
        dspsr_amd.fscrunch_fpt(ctx, torch.from_numpy(x).cuda(), out, sf)
        assert np.array_equal(out.cpu().numpy(), oracle.fscrunch_fpt(x, sf))
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.fscrunch_fpt(ctx, torch.from_numpy(x).cuda(), out, 5)     # 24 % 5


@pytest.mark.parametrize("nbit", [1, 2, 4, 8, 16, -32])
@pytest.mark.parametrize("nchan,npol,ndat,interval,flip,swap", [(1024, 1, 700, 0, True, False), (64, 2, 513, 200, False, True), (96, 4, 97, 1000, True, True)])
def test_output_stage_fpt_equals_tfp_kernels_and_oracle(oracle, gpu, nbit, nchan, npol, ndat, interval, flip, swap):
    """Rescale + SigProcDigitizer on FPT rows (Rescale.C:232-262,330-347; SigProcDigitizer.C:238-290): the bytes of the TFP kernels on
    the transposed block (same statistics, same expression per sample, same TPF byte order), the fused pass == the two operations,
    the oracle's bytes, over several blocks with Rescale intervals that end inside a block."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(43)
    ra, rb, rc = (dspsr_amd.Rescale(ctx, nchan, npol, interval, False) for _ in range(3))
    ro = oracle.Rescale(interval, False)
    nbytes = ndat * nchan * npol * (32 if nbit == -32 else nbit) // 8
    for b in range(3):
        gain = rng.uniform(0.5, 20.0, (nchan, npol, 1)).astype(np.float32) * (1 + b)
        x = (rng.standard_normal((nchan, npol, ndat)).astype(np.float32) ** 2 * gain + gain).astype(np.float32)
        x[5, 0, 3], x[6, npol - 1, 4], x[7, 0, 5] = np.inf, np.nan, 1e30
        fpt = torch.from_numpy(x).cuda()
        tfp = fpt.permute(2, 0, 1).contiguous()
        sep = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        dspsr_amd.sigproc_digitize_fpt(ctx, ra.transform_fpt(fpt.clone()), sep, nbit, True, 1.0, 0.75, flip, swap)
        ref = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        dspsr_amd.sigproc_digitize(ctx, rb.transform(tfp.clone()), ref, nchan, npol, nbit, True, 1.0, 0.75, flip, swap)
        if nbit == -32:         # floats: the channels that hold an inf / NaN are NaN in both, with whatever payload the sums left
            assert np.array_equal(sep.cpu().numpy().view(np.float32), ref.cpu().numpy().view(np.float32), equal_nan=True)
        else:
            assert torch.equal(sep, ref) and int(ref.max()) > 0
        if nbit != -32:
            fused = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            rc.digitize_fpt(fpt, fused, nbit, 0.75, flip, swap)
            assert torch.equal(fused, ref)
            (oa, sa), (oc, sc) = ra.get(), rc.get()
            assert np.array_equal(oa, oc, equal_nan=True) and np.array_equal(sa, sc, equal_nan=True)
        want = oracle.sigproc_digitize_fpt(ro.transform(x.transpose(2, 0, 1)).transpose(1, 2, 0), nbit, use_digi_scales=True, input_scale=1.0,
                                           scale_fac=0.75, flip_band=flip, swap_band=swap)
        got = ref.cpu().numpy().view(want.dtype).reshape(want.shape)
        if nbit == -32:
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6, equal_nan=True)
        else:
            d = np.abs(got.astype(np.int64) - want.astype(np.int64))     # (tree sums in double: one level, on a rounding boundary)
            assert d.max() <= 1 and (d != 0).mean() < 1e-3
    for r in (ra, rb, rc):
        r.close()


def test_fpt_kernels_beyond_the_grid_limits(oracle, gpu):
    """More rows than gridDim.y holds (70 000 channels through TScrunch / FScrunch) and more time tiles than gridDim.y holds
    (4.3 M samples through the FPT digitiser): the blocks walk on."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(50)
    x = rng.standard_normal((70000, 1, 40)).astype(np.float32)
    d_x = torch.from_numpy(x).cuda()
    out = torch.zeros((70000, 1, 5), dtype=torch.float32, device="cuda")
    carry = torch.zeros((70000, 1), dtype=torch.float32, device="cuda")
    nout, cc = dspsr_amd.tscrunch_fpt(ctx, d_x, out, 7, carry, 0)
    assert (nout, cc) == (5, 5) and np.array_equal(out.cpu().numpy(), oracle.tscrunch_fpt(x, 7))
    fo = torch.zeros((35000, 1, 40), dtype=torch.float32, device="cuda")
    dspsr_amd.fscrunch_fpt(ctx, d_x, fo, 2)
    assert np.array_equal(fo.cpu().numpy(), oracle.fscrunch_fpt(x, 2))
    nd = 65535 * 64 + 1000
    y = (rng.standard_normal((2, 1, nd)) * 3.0).astype(np.float32)
    packed = torch.zeros(nd * 2, dtype=torch.uint8, device="cuda")
    dspsr_amd.sigproc_digitize_fpt(ctx, torch.from_numpy(y).cuda(), packed, 8, use_digi_scales=True, input_scale=1.0, scale_fac=1.0)
    want = oracle.sigproc_digitize_fpt(y, 8, use_digi_scales=True, input_scale=1.0, scale_fac=1.0)
    assert np.array_equal(packed.cpu().numpy().reshape(want.shape), want)


def _fb_case(dspsr_amd, ctx, oracle, nchan, freq_res, dm, real, max_parts, input_nchan=1):
    """A filterbank object with the oracle's dedispersion kernel and its block geometry."""
    ndim = 1 if real else 2
    r = dspsr_amd.Dedispersion(1382.0, -400.0 if real else -50.0, dm, input_nchan=input_nchan, ndim=ndim)
    r.set_frequency_resolution(freq_res)
    r.match(nchan)
    fb = dspsr_amd.FilterbankEngine(ctx).setup(nchan // input_nchan, r.ndat, r.impulse_pos, r.impulse_neg, input_nchan, 2, real, r.kernel,
                                               max_parts=max_parts)
    return fb, r


@pytest.mark.parametrize("state_name", ["Intensity", "PPQQ"])
@pytest.mark.parametrize("nchan,freq_res,dm,real,sf,parts,maxp", [
    (64, 256, 0.5, True, 16, (5, 3, 1, 6), 4),          # three passes, several channels per tile, launch groups of 4 parts
    (16, 2048, 0.2, True, 7, (4, 4), 8),               # factor that divides nothing
    (128, 512, 20.0, False, 16, (6, 2, 5), 4),          # complex input (three-pass kernels on this small geometry)
    (32, 1024, 0.8, True, 1, (3, 2), 2),               # tscrunch 1: the detected samples themselves
    (8, 4096, 0.2, True, 5000, (2, 3, 1), 2),          # factor longer than a part: groups span parts and calls
    (4, 16384, 0.15, True, 16, (2, 1), 2),               # freq_res > 8192: four passes, the operations one after the other
])
def test_perform_search_equals_detection_and_tscrunch_of_the_filterbank_output(oracle, gpu, state_name, nchan, freq_res, dm, real, sf, parts, maxp):
    """Filterbank + Detection::square_law + TScrunch in one launch group (dspsr_amd_filterbank_perform_search) == the oracle's
    square_law + tscrunch_fpt applied to the SAME object's complex output, bit for bit, as a stream over several calls (the carry),
    for launch groups that split a call, Intensity and PPQQ."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(44)
    fb, r = _fb_case(dspsr_amd, ctx, oracle, nchan, freq_res, dm, real, maxp)
    npo = 1 if state_name == "Intensity" else 2
    state = dspsr_amd.INTENSITY if npo == 1 else dspsr_amd.PPQQ
    carry = torch.zeros((nchan, npo), dtype=torch.float32, device="cuda")
    cc, got, dets = 0, [], []
    for npart in parts:
        nsamp = npart * fb.nsamp_step + fb.nsamp_overlap
        raw = torch.from_numpy(np.clip(np.rint(rng.standard_normal(nsamp * 2 * (1 if real else 2)) * 24.0), -128, 127).astype(np.int8)).cuda()
        cplx = torch.zeros((nchan, 2, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
        fb.perform_raw(raw, dspsr_amd.RAW_GENERIC, 0.0123, cplx, npart)
        dets.append(oracle.square_law(cplx.cpu().numpy().view(np.complex64), state_name))
        out = torch.full((nchan, npo, (cc + npart * fb.nkeep) // sf + 1), -1.0, dtype=torch.float32, device="cuda")
        nout, cc = fb.perform_search(out, carry, cc, npart, sf, state, raw=raw, layout=dspsr_amd.RAW_GENERIC, scale=0.0123)
        got.append(out[:, :, :nout].cpu().numpy())
    want = oracle.tscrunch_fpt(np.concatenate(dets, axis=2), sf)
    got = np.concatenate(got, axis=2)
    assert got.shape == want.shape and want.shape[2] > 0
    assert np.array_equal(got, want)
    assert fb.search_is_fused() == (freq_res <= 8192)
    fb.close()


@pytest.mark.parametrize("npol,input_nchan,nchan,freq_res,dm", [
    (1, 1, 64, 512, 20.0),            # one polarisation: Intensity = Re^2 + Im^2 of it (Detection.C:218-320 with npol 1)
    (2, 4, 64, 512, 20.0),            # four input channels (a filterbank of 16 channels on each)
    (1, 2, 32, 1024, 20.0),
])
def test_perform_search_single_pol_and_multi_channel_input(oracle, gpu, npol, input_nchan, nchan, freq_res, dm):
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(49)
    r = dspsr_amd.Dedispersion(1382.0, -50.0, dm, input_nchan=input_nchan, ndim=2)
    r.set_frequency_resolution(freq_res)
    r.match(nchan)
    fb = dspsr_amd.FilterbankEngine(ctx).setup(nchan // input_nchan, r.ndat, r.impulse_pos, r.impulse_neg, input_nchan, npol, False, r.kernel,
                                               max_parts=4)
    sf = 16
    for state_name in (["Intensity", "PPQQ"] if npol == 2 else ["Intensity"]):
        npo = 1 if state_name == "Intensity" else 2
        state = dspsr_amd.INTENSITY if npo == 1 else dspsr_amd.PPQQ
        carry = torch.zeros((nchan, npo), dtype=torch.float32, device="cuda")
        cc, got, dets = 0, [], []
        for npart in (5, 3):
            nsamp = npart * fb.nsamp_step + fb.nsamp_overlap
            raw = torch.from_numpy(np.clip(np.rint(rng.standard_normal(nsamp * input_nchan * npol * 2) * 24.0), -128, 127).astype(np.int8)).cuda()
            cplx = torch.zeros((nchan, npol, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
            fb.perform_raw(raw, dspsr_amd.RAW_GENERIC, 0.0123, cplx, npart)
            c = cplx.cpu().numpy().view(np.complex64)
            if npol == 1:
                dets.append((c.real * c.real + c.imag * c.imag).astype(np.float32))
            else:
                dets.append(oracle.square_law(c, state_name))
            out = torch.full((nchan, npo, (cc + npart * fb.nkeep) // sf + 1), -1.0, dtype=torch.float32, device="cuda")
            nout, cc = fb.perform_search(out, carry, cc, npart, sf, state, raw=raw, layout=dspsr_amd.RAW_GENERIC, scale=0.0123)
            got.append(out[:, :, :nout].cpu().numpy())
        want = oracle.tscrunch_fpt(np.concatenate(dets, axis=2), sf)
        got = np.concatenate(got, axis=2)
        assert got.shape == want.shape and want.shape[2] > 0
        assert np.array_equal(got, want), (state_name, np.abs(got - want).max())
    fb.close()


def test_perform_search_real_single_pol_adds_nothing_for_the_missing_polarisation(oracle, gpu):
    """Real input with ONE polarisation: the Hermitian split of the packed transform still has a second output -- zero but for rounding,
    1e-7 of the spectrum's scale.  Intensity is Re^2 + Im^2 of the one polarisation (Detection.C:218-320), bit for bit: where a sample is
    small against that scale the other output's power moved the sum by an ulp (round 5, tests/fuzz_search.py 300 702 case 68: 2 of 4.6 M
    samples).  Random phases as the response, so the channel samples span four decades."""
    dspsr_amd, ctx = gpu
    C, M, pos, neg = 128, 4096, 1050, 19
    krng = np.random.default_rng(961203071)
    kernel = np.exp(1j * krng.uniform(-np.pi, np.pi, C * M)).astype(np.complex64)
    fb = dspsr_amd.FilterbankEngine(ctx).setup(C, M, pos, neg, 1, 1, True, kernel, max_parts=3)
    assert fb.search_is_fused()
    carry = torch.zeros((C, 1), dtype=torch.float32, device="cuda")
    cc, got, dets = 0, [], []
    for npart in (6, 5, 1):
        nsamp = npart * fb.nsamp_step + fb.nsamp_overlap
        raw = torch.from_numpy(np.clip(np.rint(krng.standard_normal(nsamp) * 24.0), -128, 127).astype(np.int8)).cuda()
        cplx = torch.zeros((C, 1, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
        fb.perform_raw(raw, dspsr_amd.RAW_GENERIC, 0.0123, cplx, npart)
        c = cplx.cpu().numpy().view(np.complex64)
        dets.append((c.real * c.real + c.imag * c.imag).astype(np.float32))
        out = torch.full((C, 1, cc + npart * fb.nkeep + 1), -1.0, dtype=torch.float32, device="cuda")
        nout, cc = fb.perform_search(out, carry, cc, npart, 1, dspsr_amd.INTENSITY, raw=raw, layout=dspsr_amd.RAW_GENERIC, scale=0.0123)
        got.append(out[:, :, :nout].cpu().numpy())
    fb.close()
    want, got = np.concatenate(dets, axis=2), np.concatenate(got, axis=2)
    assert got.shape == want.shape and np.array_equal(got, want), np.argwhere(got != want)[:4]


def test_perform_search_two_pass_geometry(oracle, gpu):
    """The two-pass path of short responses (complex dual-pol 8-bit input, -F 512:D -x 512 on a 50 MHz band: k_rows_inv with the search
    epilogue) == the three-pass kernels' complex output, detected and scrunched by the oracle."""
    dspsr_amd, ctx = gpu
    rng = np.random.default_rng(45)
    fb, r = _fb_case(dspsr_amd, ctx, oracle, 512, 512, 100.0, False, 8)
    assert fb.npass(True) == 2
    npart, sf = 12, 16
    nsamp = npart * fb.nsamp_step + fb.nsamp_overlap
    raw = torch.from_numpy(np.clip(np.rint(rng.standard_normal(nsamp * 4) * 24.0), -128, 127).astype(np.int8)).cuda()
    cplx = torch.zeros((512, 2, 2 * npart * fb.nkeep), dtype=torch.float32, device="cuda")
    fb.perform_raw(raw, dspsr_amd.RAW_GENERIC, 0.0123, cplx, npart)
    want = oracle.tscrunch_fpt(oracle.square_law(cplx.cpu().numpy().view(np.complex64), "Intensity"), sf)
    carry = torch.zeros((512, 1), dtype=torch.float32, device="cuda")
    out = torch.zeros((512, 1, want.shape[2] + 1), dtype=torch.float32, device="cuda")
    nout, cc = fb.perform_search(out, carry, 0, npart, sf, dspsr_amd.INTENSITY, raw=raw, layout=dspsr_amd.RAW_GENERIC, scale=0.0123)
    assert nout == want.shape[2] and np.array_equal(out[:, :, :nout].cpu().numpy(), want)
    fb.close()


@pytest.mark.parametrize("nbit,npol,fscrunch", [(8, 1, 0), (2, 2, 0), (8, 1, 4)])
def test_digifil_coherent_chain_against_the_oracle(oracle, gpu, nbit, npol, fscrunch):
    """LoadToFilCoherent (`digifil -F 64:D -x 1024 -t 16 -b nbit`, small enough for the float64 oracle filterbank) over three blocks:
    the packed bytes against the oracle's whole chain -- float64 filterbank of the unpacked input, then DigifilCoherent.  FFT rounding
    puts a handful of samples on the other side of a rounding boundary: at most one level apart, in well under 0.1 % of the samples."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import pipeline
    rng = np.random.default_rng(46)
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125, machine="DADA")
    cfg = pipeline.SearchConfig(nchan=64, tscrunch=16, nbit=nbit, rescale_seconds=2e-4, dispersion_measure=2.0, freq_res=1024, parts_per_block=6,
                                max_parts=4, npol=npol, fscrunch=fscrunch)
    lf = pipeline.LoadToFilCoherent(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    obs = oracle.Observation(centre_frequency=1382.0, bandwidth=-400.0, dispersion_measure=2.0)
    od = oracle.Dedispersion()
    od.set_frequency_resolution(1024)
    od.match(obs, 64)
    plan = oracle.filterbank_plan(obs, 64, od)
    assert (plan.nkeep, plan.nsamp_step) == (lf.nkeep, lf.nsamp_step)
    dig = oracle.DigifilCoherent(tscrunch=16, fscrunch=fscrunch, nbit=nbit, npol_out=npol, rescale_interval=int(2e-4 * lf.out_rate), flip_band=False)
    nblk = 6 * lf.nsamp_step + lf.nsamp_overlap
    stream = np.clip(np.rint(rng.standard_normal((3 * 6 * lf.nsamp_step + lf.nsamp_overlap) * 2) * 24.0), -128, 127).astype(np.int8)
    for b in range(3):
        raw = stream[b * 6 * lf.nsamp_step * 2:(b * 6 * lf.nsamp_step + nblk) * 2]
        got = lf.process_block(torch.from_numpy(raw.copy()).cuda()).cpu().numpy()
        fbo = oracle.filterbank(oracle.unpack_8bit(raw, obs), plan, lf.response.kernel, npart=6, dtype=np.float64).astype(np.complex64)
        want = dig.process(fbo)
        got = got.view(want.dtype).reshape(want.shape)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert want.size > 0 and d.max() <= 1 and (d != 0).mean() < 2e-3, (b, d.max(), (d != 0).mean())
    lf.close()


@pytest.mark.parametrize("nbit,npol,fscrunch,dm,freq_res,interchan", [
    (8, 4, 0, 2.0, 1024, False),        # -d 4: Coherence products with ndim 1 (LoadToFil.C:273-277)
    (8, 1, 0, 0.5, 1024, True),         # -K: SampleDelay between the filterbank and Detection (:236-247), delays of some thousand samples
    (2, 2, 2, 0.5, 1024, True),         # -K -f 2 -d 2
    (8, 1, 0, 0.0, 256, False),         # -F 64 -x 256 without :D: convolving filterbank, no response (:199-216)
    (8, 4, 0, 0.0, 0, False),           # -F 64 -d 4 with neither -x nor :D: `npol > 2` takes dsp::Filterbank with its default freq_res = 1 --
                                        # the NON-CONVOLVING filterbank (:205-207, Filterbank.C:614-623), Coherence products, ndim 1
])
def test_digifil_coherent_other_options_against_the_oracle(oracle, gpu, nbit, npol, fscrunch, dm, freq_res, interchan):
    """The same three-block comparison for the remaining options of the convolving branch."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import pipeline
    rng = np.random.default_rng(48)
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125, machine="DADA")
    ppb = 6 if freq_res else 6 * 1024                      # (freq_res = 1: one output sample per part)
    cfg = pipeline.SearchConfig(nchan=64, tscrunch=16, nbit=nbit, rescale_seconds=2e-4, dispersion_measure=dm, freq_res=freq_res, parts_per_block=ppb,
                                max_parts=4, npol=npol, fscrunch=fscrunch, dedisperse=interchan)
    lf = pipeline.LoadToFilCoherent(cfg, info, device=0, stream=torch.cuda.current_stream().cuda_stream)
    assert not lf.fused or (npol == 1 and not interchan)
    obs = oracle.Observation(centre_frequency=1382.0, bandwidth=-400.0, dispersion_measure=dm)
    od = None
    if dm:
        od = oracle.Dedispersion()
        od.set_frequency_resolution(freq_res)
        od.match(obs, 64)
    plan = oracle.filterbank_plan(obs, 64, od, freq_res=freq_res or 1)
    assert (plan.nkeep, plan.nsamp_step) == (lf.nkeep, lf.nsamp_step)
    delays = None
    if interchan:
        delays = oracle.dedispersion_sample_delays(obs, 64, lf.fb_rate)
        assert 1000 < lf.sample_delay.total_delay < ppb * lf.nkeep and lf.sample_delay.total_delay == int(max(delays) - min(delays))
    dig = oracle.DigifilCoherent(tscrunch=16, fscrunch=fscrunch, nbit=nbit, npol_out=npol, rescale_interval=int(2e-4 * lf.out_rate), flip_band=False,
                                 delays=delays)
    nblk = ppb * lf.nsamp_step + lf.nsamp_overlap
    stream = np.clip(np.rint(rng.standard_normal((3 * ppb * lf.nsamp_step + lf.nsamp_overlap) * 2) * 24.0), -128, 127).astype(np.int8)
    total = 0
    for b in range(3):
        raw = stream[b * ppb * lf.nsamp_step * 2:(b * ppb * lf.nsamp_step + nblk) * 2]
        got = lf.process_block(torch.from_numpy(raw.copy()).cuda()).cpu().numpy()
        fbo = oracle.filterbank(oracle.unpack_8bit(raw, obs), plan, lf.response.kernel, npart=ppb, dtype=np.float64).astype(np.complex64)
        want = dig.process(fbo)
        got = got.view(want.dtype).reshape(want.shape)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        if want.size:
            assert d.max() <= 1 and (d != 0).mean() < 2e-3, (b, d.max(), (d != 0).mean())
        total += want.size
    assert total > 0
    lf.close()


def test_digifil_coherent_headline_geometry_fused_equals_unfused_bytes(oracle, gpu):
    """`digifil -F 1024:D -x 4096 -D 1000 -t 16 -b 8` on the headline band (N = 2^22): the one-launch-group form (detection + time scrunch
    inside the inverse pass, Rescale + digitiser in one pass) produces the bytes of the operations run one after the other -- and of
    the oracle's DigifilCoherent fed with the SAME filterbank output (every stage behind the filterbank bit for bit)."""
    dspsr_amd, ctx = gpu
    from dspsr_amd import pipeline
    rng = np.random.default_rng(47)
    info = pipeline.InputInfo(centre_frequency=1382.0, bandwidth=-400.0, nchan=1, npol=2, ndim=1, tsamp_us=0.00125, machine="DADA")
    mk = lambda fused: pipeline.LoadToFilCoherent(pipeline.SearchConfig(nchan=1024, tscrunch=16, nbit=8, rescale_seconds=0.01, dispersion_measure=1000.0,
                                                                        freq_res=4096, parts_per_block=4, max_parts=4, fused=fused), info,
                                                  device=0, stream=torch.cuda.current_stream().cuda_stream)
    a, b = mk(True), mk(False)
    assert a.fb.search_is_fused() and a.fused and not b.fused
    dig = oracle.DigifilCoherent(tscrunch=16, nbit=8, npol_out=1, rescale_interval=int(0.01 * a.out_rate), flip_band=False)
    for blk in range(2):
        raw = torch.from_numpy(np.clip(np.rint(rng.standard_normal(a.block_bytes()) * 24.0), -128, 127).astype(np.int8)).cuda()
        pa = a.process_block(raw).clone()
        pb = b.process_block(raw).clone()
        assert pa.numel() > 0 and torch.equal(pa, pb)
        cplx = torch.zeros((1024, 2, 2 * 4 * a.nkeep), dtype=torch.float32, device="cuda")
        a.fb.perform_raw(raw, a.layout, a.scale8, cplx, 4)
        want = dig.process(cplx.cpu().numpy().view(np.complex64))
        got = pa.cpu().numpy().view(want.dtype).reshape(want.shape)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d != 0).mean() < 1e-4                # (Rescale's sums: tree in double here, sequential in the oracle)
    a.close()
    b.close()
