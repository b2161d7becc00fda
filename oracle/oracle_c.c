/* CPU ORACLE, C restatement (test infrastructure -- NOT product code).
 *
 * Plain-C restatement of the reference hot path with its OWN FFT (iterative radix-2, float
 * data, double-built twiddles), independent of numpy's pocketfft used by dspsr_oracle.py, so
 * that the two oracles pin each other's FFT conventions (forward e^{-i}, backward e^{+i}, both
 * unnormalised, as requested from cuFFT by the reference's CUDA twin,
 * Signal/General/FilterbankCUDA.cu:92,232,258).
 *
 * PARITY STATUS: "parity unpinned" at the PSRCHIVE boundary -- see dspsr_oracle.py header.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.
 *
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cf32;

/* ---------------------------------------------------------------- FFT (own, radix-2 DIT) */
static void fft_inplace(cf32 *x, unsigned n, int sign)
{
  unsigned i, j, len;
  /* bit reversal */
  for (i = 1, j = 0; i < n; i++) {
    unsigned bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { cf32 t = x[i]; x[i] = x[j]; x[j] = t; }
  }
  for (len = 2; len <= n; len <<= 1) {
    unsigned half = len >> 1;
    cf32 *tw = (cf32 *)malloc(sizeof(cf32) * half);
    for (i = 0; i < half; i++) {
      double a = sign * 2.0 * M_PI * (double)i / (double)len;
      tw[i].re = (float)cos(a);
      tw[i].im = (float)sin(a);
    }
    for (i = 0; i < n; i += len) {
      for (j = 0; j < half; j++) {
        cf32 u = x[i + j], v = x[i + j + half], w = tw[j], t;
        t.re = v.re * w.re - v.im * w.im;
        t.im = v.re * w.im + v.im * w.re;
        x[i + j].re = u.re + t.re;        x[i + j].im = u.im + t.im;
        x[i + j + half].re = u.re - t.re; x[i + j + half].im = u.im - t.im;
      }
    }
    free(tw);
  }
}

/* exported for cross-checks against numpy: sign=-1 forward, +1 backward, unnormalised */
void oracle_fft(float *data, unsigned n, int sign) { fft_inplace((cf32 *)data, n, sign); }

/* ---------------------------------------------------------------- a15: 8-bit unpack
 * value = (int8 + 0.5f) * scale      GenericEightBitUnpackerCUDA.cu:45
 * generic order ((t*nchan+c)*npol+p)*ndim+d     BitUnpacker.C:48-80
 * caspsr: 4 B pol0 / 4 B pol1        CASPSRUnpacker.C:132-187                           */
void oracle_unpack8(const int8_t *raw, uint64_t ndat, unsigned nchan, unsigned npol, unsigned ndim,
                    int caspsr, float scale, float *out /* [nchan][npol][ndat*ndim] */)
{
  uint64_t t; unsigned c, p, d;
  for (c = 0; c < nchan; c++)
    for (p = 0; p < npol; p++) {
      float *o = out + ((uint64_t)c * npol + p) * ndat * ndim;
      for (t = 0; t < ndat; t++)
        for (d = 0; d < ndim; d++) {
          uint64_t idx = caspsr ? ((t / 4) * 8 + p * 4 + (t % 4))
                                : (((t * nchan + c) * npol + p) * ndim + d);
          o[t * ndim + d] = ((float)raw[idx] + 0.5f) * scale;
        }
    }
}

/* ---------------------------------------------------------------- a2/a3: filterbank
 * Filterbank.C:561-662 (CPU branch) + Response.C:385-444 (complex multiply)
 * in : float [input_nchan][npol][ndat*ndim]; kernel: complex [input_nchan*N] or NULL
 * out: complex [nchan][npol][npart*nkeep]                                               */
void oracle_filterbank(const float *in, uint64_t in_span /* floats per (chan,pol) row */,
                       unsigned input_nchan, unsigned npol, int real_input,
                       unsigned nchan_subband, unsigned freq_res, unsigned nfilt_pos, unsigned nkeep,
                       uint64_t nsamp_step, uint64_t npart, const float *kernel, float *out)
{
  const uint64_t N = (uint64_t)nchan_subband * freq_res;
  const unsigned ndim = real_input ? 1 : 2;
  const uint64_t nfft = real_input ? 2 * N : N;
  const uint64_t out_span = npart * nkeep;            /* complex per (chan,pol) row */
  cf32 *big = (cf32 *)malloc(sizeof(cf32) * nfft);
  cf32 *small = (cf32 *)malloc(sizeof(cf32) * freq_res);
  unsigned ic, ip, s; uint64_t part, i;
  for (ic = 0; ic < input_nchan; ic++)
    for (part = 0; part < npart; part++)
      for (ip = 0; ip < npol; ip++) {
        const float *x = in + ((uint64_t)ic * npol + ip) * in_span + part * nsamp_step * ndim;
        if (real_input)       /* frc1d: 2N real -> N+1 bins, first N used (Filterbank.C:591) */
          for (i = 0; i < nfft; i++) { big[i].re = x[i]; big[i].im = 0.0f; }
        else
          for (i = 0; i < nfft; i++) { big[i].re = x[2 * i]; big[i].im = x[2 * i + 1]; }
        fft_inplace(big, (unsigned)nfft, -1);
        if (kernel) {         /* Response::operate */
          const cf32 *k = (const cf32 *)kernel + (uint64_t)ic * N;
          for (i = 0; i < N; i++) {
            float dr = big[i].re, di = big[i].im, fr = k[i].re, fi = k[i].im;
            big[i].re = fr * dr - fi * di;
            big[i].im = fi * dr + fr * di;
          }
        }
        for (s = 0; s < nchan_subband; s++) {
          cf32 *o = (cf32 *)out + (((uint64_t)ic * nchan_subband + s) * npol + ip) * out_span + part * nkeep;
          if (freq_res == 1) { o[0] = big[s]; continue; }          /* Filterbank.C:621-631 */
          memcpy(small, big + (uint64_t)s * freq_res, sizeof(cf32) * freq_res);
          fft_inplace(small, freq_res, +1);                          /* bcc1d */
          memcpy(o, small + nfilt_pos, sizeof(cf32) * nkeep);        /* :646-650 */
        }
      }
  free(big); free(small);
}

/* ---------------------------------------------------------------- a7: detection
 * cross_detect.ic:23-43 / stokes_detect.ic:21-44; p,q complex rows, outputs with stride span */
void oracle_cross_detect(unsigned ndat, const float *p, const float *q,
                         float *pp, float *qq, float *Rpq, float *Ipq, unsigned span)
{
  unsigned j;
  for (j = 0; j < ndat; j++) {
    float p_r = p[2 * j], p_i = p[2 * j + 1], q_r = q[2 * j], q_i = q[2 * j + 1];
    pp[(uint64_t)j * span] = p_r * p_r + p_i * p_i;
    qq[(uint64_t)j * span] = q_r * q_r + q_i * q_i;
    Rpq[(uint64_t)j * span] = p_r * q_r + p_i * q_i;
    Ipq[(uint64_t)j * span] = p_r * q_i - p_i * q_r;
  }
}

void oracle_stokes_detect(unsigned ndat, const float *p, const float *q,
                          float *S0, float *S1, float *S2, float *S3, unsigned span)
{
  unsigned j;
  for (j = 0; j < ndat; j++) {
    float p_r = p[2 * j], p_i = p[2 * j + 1], q_r = q[2 * j], q_i = q[2 * j + 1];
    float pp = p_r * p_r + p_i * p_i, qq = q_r * q_r + q_i * q_i;
    S0[(uint64_t)j * span] = pp + qq;
    S1[(uint64_t)j * span] = pp - qq;
    S2[(uint64_t)j * span] = (float)(2.0 * (p_r * q_r + p_i * q_i));
    S3[(uint64_t)j * span] = (float)(2.0 * (p_r * q_i - p_i * q_r));
  }
}

/* ---------------------------------------------------------------- a10: bin plan
 * Fold.C:744-787 sequential double recurrence; returns hits[] increments too            */
void oracle_fold_binplan(double phi, double phase_per_sample, unsigned nbin, uint64_t ndat_fold,
                         unsigned *binplan, unsigned *hits)
{
  uint64_t i; const double double_nbin = (double)nbin;
  for (i = 0; i < ndat_fold; i++) {
    unsigned ibin;
    phi -= floor(phi);
    ibin = (unsigned)(phi * double_nbin);
    phi += phase_per_sample;
    binplan[i] = ibin;
    if (hits) hits[ibin]++;
  }
}

/* ---------------------------------------------------------------- a11: fold accumulate
 * Fold.C:835-872 (FPT order): profile[chan][pol][bin][dim] += x[chan][pol][idat][dim]   */
void oracle_fold(const float *in, uint64_t in_span /* floats per (chan,pol) row */,
                 unsigned nchan, unsigned npol, unsigned ndim, uint64_t idat_start, uint64_t ndat_fold,
                 const unsigned *binplan, unsigned nbin, float *profile /* [nchan][npol][nbin][ndim] */)
{
  unsigned c, p, d; uint64_t i;
  for (c = 0; c < nchan; c++)
    for (p = 0; p < npol; p++) {
      const float *timep = in + ((uint64_t)c * npol + p) * in_span + idat_start * ndim;
      float *phasep = profile + ((uint64_t)c * npol + p) * nbin * ndim;
      for (i = 0; i < ndat_fold; i++) {
        if (binplan[i] != nbin) {
          float *ph = phasep + (uint64_t)binplan[i] * ndim;
          for (d = 0; d < ndim; d++) ph[d] += timep[d];
        }
        timep += ndim;
      }
    }
}
