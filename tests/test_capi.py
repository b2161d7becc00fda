"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dspsr_amd.h declares; the host-side preparation entry points (which need no GPU) agree with
the oracle; device entry points fail cleanly (error code, no crash) when no HIP device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "dspsr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dspsr_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import dspsr_amd
    names = _declared_functions()
    assert len(names) >= 30
    lib = C.CDLL(dspsr_amd.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/dspsr_amd.h is not exported" % n
    # and the Python binding table covers the same set
    from dspsr_amd import _lib
    assert sorted(_lib.SYMBOLS) == names


def test_shipped_library_matches_the_sources():
    """dspsr_amd_build_id() = sha256 of the sources the library was built from (csrc/Makefile: BUILD_ID), recomputed here from the
    tree: a stale libdspsr_amd.so -- a source edited without `make` -- fails the CPU suite, and the evidence files under profiles/
    that carry the id can be matched to a library."""
    import hashlib
    import dspsr_amd
    from dspsr_amd import _lib
    csrc = os.path.join(ROOT, "dspsr_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    units = re.search(r"^UNITS = (.*)$", mk, re.M).group(1).split()
    hdr = re.search(r"^HDR = (.*)$", mk, re.M).group(1).split()
    files = sorted([u + ".hip" for u in units] + ["host_prep.cpp"] + hdr) + ["Makefile"]
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    assert _lib.lib.dspsr_amd_build_id().decode() == h.hexdigest()[:12], "libdspsr_amd.so is older than its sources: run make -C dspsr_amd/csrc"
    assert h.hexdigest()[:12] in _lib.lib.dspsr_amd_version().decode()
    assert dspsr_amd.build_id() == h.hexdigest()[:12]


def test_shipped_library_reads_no_environment():
    """The library's behaviour depends on the configuration structs of the C-ABI alone: it neither imports getenv nor holds any
    of the knob names earlier rounds' experiment builds read; the phase-stamp diagnostics (csrc/stamps.h) are not in it."""
    import subprocess
    import dspsr_amd
    blob = open(dspsr_amd.LIB_PATH, "rb").read()
    assert b"DSPSR_AMD_DEBUG" not in blob and b"DSPSR_AMD_FUSED_MIN_TILES" not in blob and b"DSPSR_AMD_WG_PER_CU" not in blob
    undefined = subprocess.run(["nm", "-D", "--undefined-only", dspsr_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    exported = subprocess.run(["nm", "-D", "--defined-only", dspsr_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "debug_stamps" not in exported


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from dspsr_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import dspsr_amd
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.Context(0)


@pytest.mark.parametrize("f0,bw,dm,nchan,x", [
    (1382, -400, 1000, 1024, 4096), (1382, -400, 2000, 1024, 4096), (2000, -400, 500, 256, 4096),
    (1382, -400, 67.99, 64, 0), (1400, 64, 10, 16, 0), (1382, -50, 1000, 512, 0)])
def test_dedispersion_prepare_and_build_match_oracle(oracle, f0, bw, dm, nchan, x):
    import dspsr_amd
    d = dspsr_amd.Dedispersion(f0, bw, dm)
    if x:
        d.set_frequency_resolution(x)
    d.match(nchan)
    obs = oracle.Observation(centre_frequency=f0, bandwidth=bw, dispersion_measure=dm)
    r = oracle.Dedispersion()
    if x:
        r.set_frequency_resolution(x)
    r.match(obs, nchan)
    assert (d.impulse_pos, d.impulse_neg, d.ndat, d.minimum_ndat) == \
        (r.impulse_pos, r.impulse_neg, r.ndat, r.get_minimum_ndat())
    # phases are rounded to float identically; cosf/sinf vs double cos/sin rounded may differ by 1 ulp
    assert np.abs(d.kernel - r.buffer).max() <= 1.2e-7
    assert d.kernel[0] == 0


@pytest.mark.parametrize("in_nchan,ndim,dsb", [(1, 2, -1), (8, 2, -1), (8, 2, 0), (1, 2, 0)])
def test_dedispersion_swap_ordering_matches_oracle(oracle, in_nchan, ndim, dsb):
    """Response::match band-swap bookkeeping for complex (dual-sideband) input, Response.C:132-181."""
    import dspsr_amd
    f0, bw, dm, nchan = 1382.0, -400.0, 2.0, 64
    d = dspsr_amd.Dedispersion(f0, bw, dm, input_nchan=in_nchan, ndim=ndim, dual_sideband=dsb)
    d.set_frequency_resolution(1024)
    d.match(nchan)
    obs = oracle.Observation(centre_frequency=f0, bandwidth=bw, dispersion_measure=dm, nchan=in_nchan, ndim=ndim,
                             dual_sideband=dsb)
    r = oracle.Dedispersion()
    r.set_frequency_resolution(1024)
    r.match(obs, nchan)
    assert np.abs(d.kernel - r.buffer).max() <= 1.2e-7


def test_check_ndat_error_text():
    import dspsr_amd
    d = dspsr_amd.Dedispersion(1382, -400, 500)
    d.set_frequency_resolution(4096)
    with pytest.raises(dspsr_amd.DspsrAmdError, match=r"specified ndat \(4096\) < required minimum ndat \(8192\)"):
        d.match(256)


@pytest.mark.parametrize("nbad", [1, 3, 54, 844, 1687, 1909, 6735, 14567])
def test_optimal_fft_length_matches_oracle(oracle, nbad):
    import dspsr_amd
    for nmax in (0, 1 << 16):
        assert dspsr_amd.optimal_fft_length(nbad, nmax) == oracle.optimal_fft_length(nbad, nmax)


def test_eight_bit_scale_and_binplan(oracle):
    import dspsr_amd
    assert dspsr_amd.eight_bit_scale() == float(oracle.S8)
    plan, hits = dspsr_amd.fold_binplan(0.731, 1.0 / 345.67, 1024, 100000)
    want = oracle.fold_binplan(0.731, 1.0 / 345.67, 1024, 100000)
    assert np.array_equal(plan, want)                      # double recurrence, bit exact
    assert np.array_equal(hits, np.bincount(want, minlength=1024))


def test_golden_ref_kat_through_product_host_code(oracle):
    """optimal_fft_length table produced by the reference's own optimize_fft.c (tests/golden/ref_kat.npz)."""
    import dspsr_amd
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_kat.npz"))
    for nbad, nopt in zip(fx["nbad"], fx["nopt"]):
        assert dspsr_amd.optimal_fft_length(int(nbad), 0) == int(nopt)
    # and the oracle's detection against the reference's cross_detect.c / stokes_detect.c outputs
    fb = np.stack([fx["p"].view(np.complex64), fx["q"].view(np.complex64)])[None]
    assert np.array_equal(oracle.detect_products(fb, "Coherence")[0], fx["cross"])
    assert np.array_equal(oracle.detect_products(fb, "Stokes")[0], fx["stokes"])


def test_fractional_delay_kernel_matches_oracle(oracle):
    """-K: the fractional inter-channel delay phase of Dedispersion::build (Dedispersion.C:524-545) in the product's
    host code against the oracle, and a sanity check that it is a pure linear phase per channel."""
    import dspsr_amd
    o = oracle
    f0, bw, dm, nchan = 1382.0, -64.0, 10.0, 16
    d = dspsr_amd.Dedispersion(f0, bw, dm, fractional_delay=True)
    d.set_frequency_resolution(1024)
    d.match(nchan)
    obs = o.Observation(centre_frequency=f0, bandwidth=bw, dispersion_measure=dm)
    od = o.Dedispersion()
    od.fractional_delay = True
    od.set_frequency_resolution(1024)
    od.match(obs, nchan)
    assert np.abs(d.kernel - od.buffer).max() <= 1.2e-7          # cosf/sinf vs rounded double cos/sin: 1 ulp
    plain = dspsr_amd.Dedispersion(f0, bw, dm)
    plain.set_frequency_resolution(1024)
    plain.match(nchan)
    ratio = (d.kernel.reshape(nchan, -1)[:, 1:] * np.conj(plain.kernel.reshape(nchan, -1)[:, 1:]))
    ph = np.unwrap(np.angle(ratio.astype(np.complex128)), axis=1)
    slope = np.diff(ph, axis=1)
    assert np.abs(slope - slope.mean(axis=1, keepdims=True)).max() < 2e-3       # float-phase quantisation only
    assert np.abs(slope.mean(axis=1)).max() > 0                                  # and it is not the identity


@pytest.mark.parametrize("f0,bw,dm,nchan,swap,nsub,dc", [(1382.0, -400.0, 71.0, 1024, False, 0, False), (1400.0, 64.0, 10.0, 16, True, 0, False),
                                                         (2000.0, -400.0, 500.0, 256, False, 4, True), (1382.0, -64.0, 3.0, 1, False, 0, False)])
def test_dedispersion_sample_delays_match_oracle(oracle, f0, bw, dm, nchan, swap, nsub, dc):
    """-K integer delays: DedispersionSampleDelay.C:24-75 in the product's host code vs the oracle (double, exact)."""
    import dspsr_amd
    rate = abs(bw) * 1e6 / nchan
    obs = oracle.Observation(centre_frequency=f0, bandwidth=bw, dispersion_measure=dm, dc_centred=dc)
    got = dspsr_amd.dedispersion_sample_delays(f0, bw, dm, nchan, rate, swap, nsub, dc)
    want = oracle.dedispersion_sample_delays(obs, nchan, rate, swap, nsub)
    assert np.array_equal(got, want)
    if nchan > 1:
        assert got.max() > 0 > got.min()                     # relative to the centre frequency
    with pytest.raises(dspsr_amd.DspsrAmdError):
        dspsr_amd.dedispersion_sample_delays(f0, 0.0, dm, nchan, rate)


def test_zero_dm_needs_an_explicit_resolution(oracle):
    """DM = 0: no smearing, optimal_fft_length(0) fails and the reference throws (Response.C:300-305); same here, and -x works."""
    import dspsr_amd
    r = dspsr_amd.Dedispersion(1382.0, -16.0, 0.0, input_nchan=1, ndim=1)
    with pytest.raises(dspsr_amd.DspsrAmdError, match="Response::set_optimal_ndat optimal_fft_length failed"):
        r.match(16)
    obs = oracle.Observation(centre_frequency=1382.0, bandwidth=-16.0, tsamp_us=1 / 32, dispersion_measure=0.0)
    with pytest.raises(oracle.OracleError, match="optimal_fft_length failed"):
        oracle.Dedispersion().match(obs, 16)
    r.set_frequency_resolution(64)
    r.match(16)
    assert (r.ndat, r.impulse_pos, r.impulse_neg) == (64, 0, 0)


def test_host_preparation_random_bands_against_the_oracle(oracle):
    """Random bands / DMs / channel counts / -x: dspsr_amd_dedispersion_prepare + build agree with the restated
    Dedispersion::prepare / build / match (impulse_pos, impulse_neg, ndat, kernel), and refuse exactly where it refuses
    (smearing beyond the threshold, -x below the minimum, DM 0 without -x)."""
    import dspsr_amd
    rng = np.random.default_rng(5)
    agree = refuse = 0
    for _ in range(24):
        freq = float(rng.choice([150.0, 400.0, 1382.0, 3100.0]))
        bw = float(rng.choice([-64, -16, 8, 32]))
        dm = float(rng.choice([0.0, 0.5, 3, 30, 300]))
        in_nchan = int(rng.choice([1, 1, 2, 8]))
        ndim = 1 if in_nchan == 1 and rng.integers(0, 2) else 2
        nchan = in_nchan * int(2 ** rng.integers(0, 5))
        fr = int(rng.choice([0, 0, 64, 4096]))
        obs = oracle.Observation(centre_frequency=freq, bandwidth=bw, nchan=in_nchan, npol=2, ndim=ndim,
                                 tsamp_us=(0.5 if ndim == 1 else 1.0) / abs(bw / in_nchan), dispersion_measure=dm)
        rp = dspsr_amd.Dedispersion(freq, bw, dm, input_nchan=in_nchan, ndim=ndim)
        ro = oracle.Dedispersion()
        if fr:
            rp.set_frequency_resolution(fr)
            ro.set_frequency_resolution(fr)
        ep = eo = None
        try:
            rp.match(nchan)
        except dspsr_amd.DspsrAmdError as e:
            ep = str(e)
        if ep is None and rp.kernel.size > (1 << 20):
            continue                                        # (the Python restatement builds big kernels slowly)
        try:
            ro.match(obs, nchan)
        except oracle.OracleError as e:
            eo = str(e)
        assert (ep is None) == (eo is None), (freq, bw, dm, in_nchan, ndim, nchan, fr, ep, eo)
        if ep is None:
            assert (rp.impulse_pos, rp.impulse_neg, rp.ndat) == (ro.impulse_pos, ro.impulse_neg, ro.ndat)
            assert np.abs(rp.kernel - ro.buffer).max() <= 2e-7
            agree += 1
        else:
            refuse += 1
    assert agree >= 8 and refuse >= 2
