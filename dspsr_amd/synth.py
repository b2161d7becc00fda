"""Synthetic DADA input: seeded Gaussian noise plus a dispersed, pulsed noise component.

The reference ships no signal generator (its benchmarks run on all-zero DUMMY data,
Kernel/Classes/DummyFile.C:78-84); this one exists so that parity tests fold something
non-trivial.  Format facts: 4096-byte ASCII header + two's-complement int8 samples in the
generic order ((t*nchan+c)*npol+p)*ndim+d (Kernel/Classes/BitUnpacker.C:48-80) or the CASPSR
4-sample pol interleave (Kernel/Formats/caspsr/CASPSRUnpacker.C:132-187)."""
from __future__ import annotations

import numpy as np

SEED = 20100413
DM_DISPERSION = 2.41e-4


def dada_header(freq, bw, nchan, npol, ndim, tsamp_us, instrument="DADA", source="J0835-4510",
                utc_start="2010-04-13-02:05:45", size=4096, extra=None) -> bytes:
    lines = ["HDR_VERSION 1.0", "HDR_SIZE %d" % size, "BW %s" % repr(float(bw)), "FREQ %s" % repr(float(freq)),
             "TELESCOPE PKS", "RECEIVER MULTI", "INSTRUMENT %s" % instrument, "SOURCE %s" % source, "MODE PSR",
             "NBIT 8", "NCHAN %d" % nchan, "NDIM %d" % ndim, "NPOL %d" % npol, "OBS_OFFSET 0",
             "UTC_START %s" % utc_start, "TSAMP %s" % repr(float(tsamp_us))]
    for k, v in (extra or {}).items():
        lines.append("%s %s" % (k, v))
    raw = ("\n".join(lines) + "\n").encode("ascii")
    assert len(raw) <= size
    return raw + b"\0" * (size - len(raw))


def _pulse_envelope(t_sec, period, duty=0.05, amp=1.0):
    ph = (t_sec / period) % 1.0
    return 1.0 + amp * np.exp(-0.5 * ((ph - 0.5) / (duty / 2.355)) ** 2)


def voltages(ndat, freq, bw, tsamp_us, dm, period, npol=2, ndim=1, nchan=1, sigma=24.0, pulse_amp=3.0,
             seed=SEED, layout="generic") -> np.ndarray:
    """int8 raw bytes of `ndat` time samples.  Pulse-modulated noise is dispersed (circularly,
    over the whole array) with the inverse of the coherent-dedispersion chirp of the full band."""
    rng = np.random.default_rng(seed)
    out = np.empty((ndat, nchan, npol, ndim), dtype=np.int8)
    t = np.arange(ndat) * (tsamp_us * 1e-6)
    env = np.sqrt(_pulse_envelope(t, period, amp=pulse_amp))
    disp = 1e6 * dm / DM_DISPERSION
    chbw = bw / nchan
    for c in range(nchan):
        # channel centre as dsp::Observation::get_centre_frequency(ichan) orders them
        fc = freq - 0.5 * bw + (c + 0.5) * chbw
        for p in range(npol):
            if ndim == 1:
                x = rng.standard_normal(ndat) * env
                X = np.fft.rfft(x)
                # real-sampled band (sampling rate 2 |bw|): rfft bin k of ndat samples lies k * 2 bw / ndat from the band edge at DC,
                # i.e. at sky frequency fc - bw/2 + 2 k bw / ndat  (round 5: the factor 2 was missing -- the band was dispersed as
                # if it were half as wide, a quarter of the delay per MHz; parity tests compare product and oracle on the same
                # input and did not notice, the physical test tests/test_dedispersion_physics.py does)
                f = (np.arange(X.size) / ndat) * (2.0 * chbw) - 0.5 * chbw
            else:
                x = (rng.standard_normal(ndat) + 1j * rng.standard_normal(ndat)) * env * np.sqrt(0.5)
                X = np.fft.fft(x)
                f = np.fft.fftfreq(ndat) * chbw
            sign = 1.0 if chbw > 0 else -1.0
            phase = -sign * 2 * np.pi * disp / (fc * fc) * (f * f) / (fc + f)   # dedispersion phase
            X = X * np.exp(-1j * phase)                                          # dispersion = inverse
            y = np.fft.irfft(X, ndat) if ndim == 1 else np.fft.ifft(X)
            if ndim == 1:
                q = np.clip(np.rint(y * sigma - 0.5), -128, 127).astype(np.int8)
                out[:, c, p, 0] = q
            else:
                out[:, c, p, 0] = np.clip(np.rint(y.real * sigma * np.sqrt(2) - 0.5), -128, 127).astype(np.int8)
                out[:, c, p, 1] = np.clip(np.rint(y.imag * sigma * np.sqrt(2) - 0.5), -128, 127).astype(np.int8)
    raw = out.reshape(-1)
    if layout == "caspsr":
        assert nchan == 1 and npol == 2 and ndim == 1 and ndat % 4 == 0
        raw = out.reshape(ndat // 4, 4, 2).transpose(0, 2, 1).reshape(-1)
    return np.ascontiguousarray(raw)
