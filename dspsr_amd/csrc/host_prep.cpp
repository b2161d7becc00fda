// Host-side preparation that the reference also performs on the host:
//   dsp::Dedispersion   Signal/General/Dedispersion.C:216-248 (prepare), :291-331 (build),
//                       :383-475 (smearing), :478-556 (phases)
//   dsp::Response       Signal/General/Response.C:132-181 (match ordering), :259-344 (ndat rules),
//                       :649-700 (doswap);  dsp::Shape::rotate  Shape.C:222-266
//   optimal_fft_length  Signal/General/optimize_fft.c:63-127
//   BitTable scale      Kernel/Classes/BitTable.C:165-218
//   Fold bin plan       Signal/Pulsar/Fold.C:744-787
// The kernel is built in double, phases are rounded to float exactly where the reference rounds
// them (vector<float> phases), so the device receives the same numbers the CUDA engine would.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <complex>
#include <vector>

#include "../../include/dspsr_amd.h"

namespace {

const double dm_dispersion = 2.41e-4;                      // Dedispersion.C:28
const double smearing_buffer = 0.1;                        // Dedispersion.C:30
const unsigned smearing_samples_threshold = 16 * 1024 * 1024;   // Dedispersion.C:214

struct Dedisp {
  double centre_frequency, bandwidth, dm;
  unsigned nchan;
  std::vector<bool> supported;

  double delay_time(double f1, double f2) const            // :348-356
  {
    const double dispersion = dm / dm_dispersion;
    return dispersion * (1.0 / (f1 * f1) - 1.0 / (f2 * f2));
  }
  double smearing_time(int half) const                     // :383-430
  {
    const double abs_bw = fabs(bandwidth);
    double ch_abs_bw = abs_bw / double(nchan);
    double lower_ch_cfreq = centre_frequency - (abs_bw - ch_abs_bw) / 2.0;
    unsigned ichan = 0;
    while (ichan < supported.size() && !supported[ichan]) { lower_ch_cfreq += ch_abs_bw; ichan++; }
    if (half) { ch_abs_bw /= 2.0; lower_ch_cfreq += double(half) * ch_abs_bw; }
    return delay_time(lower_ch_cfreq - fabs(0.5 * ch_abs_bw), lower_ch_cfreq + fabs(0.5 * ch_abs_bw));
  }
  unsigned smearing_samples(int half) const                // :432-475
  {
    double tsmear = smearing_time(half);
    const double sampling_rate = fabs(bandwidth) / double(nchan) * 1e6;
    tsmear *= (1.0 + smearing_buffer);
    return unsigned(ceil(tsmear * sampling_rate));
  }
};

unsigned minimum_ndat(unsigned pos, unsigned neg)          // Response.C:259-275
{
  const double impulse_tot = pos + neg;
  if (impulse_tot == 0) return 0;
  unsigned min = unsigned(pow(2.0, ceil(log(impulse_tot) / log(2.0))));
  while (min <= impulse_tot) min *= 2;
  return min;
}

void swap_halves(std::complex<float>* buf, uint64_t npts, unsigned divisions)   // Response.C:649-700
{
  const uint64_t half = npts / (2 * divisions);
  for (unsigned d = 0; d < divisions; d++) {
    std::complex<float>* p1 = buf + 2 * half * d;
    std::swap_ranges(p1, p1 + half, p1 + half);
  }
}

bool get_dual_sideband(const dspsr_amd_dedispersion_config* c)   // Observation.C:80-87
{
  if (c->dual_sideband != -1) return c->dual_sideband == 1;
  return c->ndim == 2;
}

int fail(char* errbuf, size_t errlen, const char* fmt, unsigned a, unsigned b)
{
  if (errbuf && errlen) snprintf(errbuf, errlen, fmt, a, b);
  return DSPSR_AMD_EINVAL;
}

}  // namespace

extern "C" uint64_t dspsr_amd_optimal_fft_length(uint64_t nbadperfft, uint64_t nfft_max)   // optimize_fft.c:63-127
{
  if (!nbadperfft) return (uint64_t)-1;
  uint64_t nfft_min = (uint64_t)pow(2.0, ceil(log((double)nbadperfft) / log(2.0)));
  if (nfft_max && nfft_max < nfft_min) return (uint64_t)-1;
  uint64_t nfft = nfft_min;
  double timescale = ((double)nfft * log((double)nfft)) / (double)(nfft - nbadperfft);
  while (nfft_max == 0 || nfft * 2 < nfft_max) {
    const double prev = timescale;
    nfft *= 2;
    timescale = ((double)nfft * log((double)nfft)) / (double)(nfft - nbadperfft);
    if (timescale > prev) { nfft /= 2; break; }
  }
  return nfft;
}

extern "C" int dspsr_amd_dedispersion_prepare(const dspsr_amd_dedispersion_config* cfg,
                                              dspsr_amd_dedispersion_info* info, char* errbuf, size_t errlen)
{
  if (!cfg || !info || !cfg->nchan || cfg->bandwidth == 0.0) return DSPSR_AMD_EINVAL;
  Dedisp d;
  d.centre_frequency = cfg->centre_frequency;
  d.bandwidth = cfg->bandwidth;
  d.dm = cfg->dispersion_measure;
  d.nchan = cfg->nchan;
  d.supported.assign(cfg->nchan, true);
  const unsigned threshold = smearing_samples_threshold / cfg->nchan;        // Dedispersion.C:216-240
  unsigned ichan = 0, neg;
  while ((neg = d.smearing_samples(-1)) > threshold) {
    d.supported[ichan] = false;
    ichan++;
    if (ichan == cfg->nchan)
      return fail(errbuf, errlen, "dsp::Dedispersion::prepare smearing samples=%u exceeds threshold=%u", neg, threshold);
  }
  info->impulse_neg = neg;
  info->impulse_pos = d.smearing_samples(1);
  info->minimum_ndat = minimum_ndat(info->impulse_pos, info->impulse_neg);
  if (cfg->freq_res) {                                                          // Response::check_ndat :328-344
    if (cfg->ndat_max && cfg->freq_res > cfg->ndat_max)
      return fail(errbuf, errlen, "Response::check_ndat specified maximum ndat (%d) < specified ndat (%d)",
                  cfg->ndat_max, cfg->freq_res);
    if (cfg->freq_res < info->minimum_ndat)
      return fail(errbuf, errlen, "dsp::Response::check_ndat specified ndat (%d) < required minimum ndat (%d)",
                  cfg->freq_res, info->minimum_ndat);
    info->ndat = cfg->freq_res;
  } else {                                                                      // set_optimal_ndat :282-311
    if (cfg->ndat_max && cfg->ndat_max < info->minimum_ndat)
      return fail(errbuf, errlen,
                  "Response::set_optimal_ndat specified maximum ndat (%d) < required minimum ndat (%d)",
                  cfg->ndat_max, info->minimum_ndat);
    // (DM = 0: no smearing, nbadperfft == 0 -- optimal_fft_length fails and the reference throws, Response.C:300-305;
    //  an explicit resolution, -x, is the way through there as here)
    const uint64_t n = dspsr_amd_optimal_fft_length(info->impulse_pos + info->impulse_neg, cfg->ndat_max);
    if (n == (uint64_t)-1) return fail(errbuf, errlen, "Response::set_optimal_ndat optimal_fft_length failed%s", 0, 0);
    info->ndat = (uint32_t)n;
  }
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_dedispersion_build(const dspsr_amd_dedispersion_config* cfg, uint32_t ndat,
                                            float* kernel_host)
{
  if (!cfg || !kernel_host || !ndat || !cfg->nchan || cfg->bandwidth == 0.0) return DSPSR_AMD_EINVAL;
  const unsigned nchan = cfg->nchan;
  std::complex<float>* phasors = reinterpret_cast<std::complex<float>*>(kernel_host);

  // Dedispersion::build(vector<float>&, ndat, nchan)  :478-556  (Doppler_shift = 1)
  const double centrefreq = cfg->centre_frequency, bw = cfg->bandwidth;
  const double sign = bw / fabs(bw);
  const double chanwidth = bw / double(nchan);
  const double binwidth = chanwidth / double(ndat);
  double lower_cfreq = centrefreq - 0.5 * bw;
  if (!cfg->dc_centred) lower_cfreq += 0.5 * chanwidth;
  const double dispersion_per_MHz = 1e6 * cfg->dispersion_measure / dm_dispersion;
  const double highest_freq = centrefreq + 0.5 * fabs(bw - chanwidth);            // :504
  const double samp_int = 1.0 / chanwidth;                                         // :506 (microseconds, signed like bw)
  for (unsigned ichan = 0; ichan < nchan; ichan++) {
    const double chan_cfreq = lower_cfreq + double(ichan) * chanwidth;
    double delay = 0.0;
    if (cfg->fractional_delay) {                                                   // :524-533
      delay = dispersion_per_MHz * (1.0 / (chan_cfreq * chan_cfreq) - 1.0 / (highest_freq * highest_freq));
      delay = -fmod(delay, samp_int);
    }
    const double coeff = -sign * 2 * M_PI * dispersion_per_MHz / (chan_cfreq * chan_cfreq);
    const uint64_t spt = (uint64_t)ichan * ndat;
    for (unsigned ipt = 0; ipt < ndat; ipt++) {
      const double freq = double(ipt) * binwidth - 0.5 * chanwidth;
      const double delay_phase = -2.0 * M_PI * freq * delay;                       // :543
      const float phase = float(coeff * (freq * freq) / (chan_cfreq + freq) + delay_phase);   // stored as float :545
      phasors[spt + ipt] = std::polar(float(1.0), phase);                       // :320
    }
  }
  phasors[0] = 0;                                                                // :323

  // Response::match(const Observation*, unsigned)  Response.C:132-181
  const uint64_t npts = (uint64_t)nchan * ndat;
  if (cfg->input_nchan == 1) {
    if (get_dual_sideband(cfg)) swap_halves(phasors, npts, 1);
  } else {
    if (cfg->dc_centred) {
      // Dedispersion::prepare copied dc_centred from the input, so Response::dc_centred is already
      // true here and the half-channel rotate (Shape.C:222-266) is skipped -- same as the reference.
    }
    if (get_dual_sideband(cfg)) swap_halves(phasors, npts, cfg->input_nchan);
    if (cfg->swap) swap_halves(phasors, npts, 1);
  }
  phasors[0] = 0;                                                                // Dedispersion.C:278
  return DSPSR_AMD_OK;
}

// Dedispersion::SampleDelay::match (DedispersionSampleDelay.C:24-75) with Observation::get_centre_frequency(ichan)
// (Kernel/Classes/Observation.C:420-451)
extern "C" int dspsr_amd_dedispersion_sample_delays(double centre_frequency, double bandwidth, double dispersion_measure,
                                                    uint32_t nchan, double rate_hz, int swap, uint32_t nsub_swap,
                                                    int dc_centred, int64_t* delays_host)
{
  if (!delays_host || !nchan) return DSPSR_AMD_EINVAL;
  if (rate_hz == 0 || bandwidth == 0 || centre_frequency == 0) return DSPSR_AMD_EINVAL;   // :50-55 "invalid input"
  const double dispersion = dispersion_measure / 2.41e-4;                                 // dm_dispersion, Dedispersion.C:28
  const double base = dc_centred ? centre_frequency - 0.5 * bandwidth
                                 : centre_frequency - 0.5 * bandwidth + 0.5 * bandwidth / double(nchan);
  for (uint32_t ichan = 0; ichan < nchan; ichan++) {
    uint32_t c = ichan;
    if (swap) c = (c + nchan / 2) % nchan;
    if (nsub_swap) {
      const uint32_t sub_nchan = nchan / nsub_swap;
      c = (c / sub_nchan) * sub_nchan + (c % sub_nchan + sub_nchan / 2) % sub_nchan;
    }
    const double freq = base + double(c) * bandwidth / double(nchan);
    const double delay = dispersion * (1.0 / (centre_frequency * centre_frequency) - 1.0 / (freq * freq));
    delays_host[ichan] = int64_t(floor(delay * rate_hz + 0.5));
  }
  return DSPSR_AMD_OK;
}

extern "C" double dspsr_amd_eight_bit_scale(double input_spacing)               // BitTable.C:165-218
{
  const unsigned unique_values = 256;
  const double output_spacing = 1.0 / double(unique_values);
  const double output_middle = double(unique_values - 1) / 2.0;
  const unsigned input_middle = unique_values / 2;
  double cumulative_probability = 0.0, variance = 0.0;
  for (unsigned i = 0; i < unique_values; i++) {
    const double output = (double(i) - output_middle) * output_spacing;
    if (i < input_middle) {
      const double threshold = double(int(i + 1) - int(input_middle)) * input_spacing;
      const double cumulative = 0.5 * (1.0 + erf(threshold / sqrt(2.0)));   // NormalDistribution (ext)
      const double interval = cumulative - cumulative_probability;
      cumulative_probability = cumulative;
      variance += output * output * interval;
    }
  }
  variance *= 2.0;
  return (1.0 / sqrt(variance)) * output_spacing;
}

// One RUN of the plan loop of Fold.C:744-787 without walking its samples.
//   per sample:  if (!(phi >= 0 && phi < 1)) phi -= floor(phi);  ibin = unsigned(phi * nbin);  phi += phase_per_sample;
// The recurrence is a chain of rounded additions, so its values cannot be had by a multiplication -- but inside one binade
// [2^(e-1), 2^e) every phi is a multiple K u of the spacing u = 2^(e-53), and the rounded sum phi + pps is (K + round(pps / u)) u for
// every K as long as it stays in the binade and pps / u is not half-way between two integers: the chain is an exact arithmetic
// progression there, in integers.  The bin number is monotone along it, so the last sample of the run is found by bisection with the
// reference's own expression unsigned(phi * nbin).  A step that leaves the binade (or an undecided case: tie, phi tiny, pps < 0) is
// taken as the one rounded addition it is.  Same values as the sample-by-sample loop, bit for bit (tests/test_fold_plan_runs.py:
// against dspsr_amd_fold_binplan on random phases, steps and bin counts); cost per run instead of per sample -- at 50 MHz channel
// rates (`dspsr -F 8`) the loop was twice the time of the kernels of a block.
// In: *phi the state in front of the first sample (not normalised), nmax >= 1 samples available.  Out: the bin of the run, the
// number of its samples taken (<= nmax), *phi the state in front of the sample behind them.
namespace dspsr_amd {
static inline double pow2d(const int e)                      // 2^e for -1022 <= e <= 1023
{
  const uint64_t b = (uint64_t)(e + 1023) << 52;
  double d;
  memcpy(&d, &b, sizeof d);
  return d;
}
uint64_t fold_plan_run(double* phi_io, const double pps, const double double_nbin, const uint64_t nmax, uint32_t* ibin_out)
{
  double phi = *phi_io;
  if (!(phi >= 0.0 && phi < 1.0)) phi -= floor(phi);
  const uint32_t ibin = (uint32_t)(phi * double_nbin);
  *ibin_out = ibin;
  uint64_t n = 0;
  for (;;) {
    // sample n of the run has phase phi (in [0, 1), unless the input was not finite) and lies in bin ibin
    bool stepped = false;
    if (pps >= 0.0 && phi >= 0x1p-900 && phi < 1.0) {
      uint64_t bits;
      memcpy(&bits, &phi, sizeof bits);
      const int e = (int)((bits >> 52) & 0x7ff) - 1022;        // phi in [2^(e-1), 2^e) (normal: phi >= 2^-900), spacing u = 2^(e-53)
      const double r = pps * pow2d(53 - e);                    // pps / u: a power-of-two scaling, exact or inf (53 - e <= 953)
      if (r < 0x1p53) {
        const double fl = floor(r), frac = r - fl;
        if (frac != 0.5) {
          const uint64_t Ci = (uint64_t)(frac > 0.5 ? fl + 1.0 : fl);
          const uint64_t K = (bits & 0xfffffffffffffull) | (1ull << 52);      // phi / u: 2^52 <= K < 2^53
          const uint64_t kcap = Ci ? ((1ull << 53) - 1 - K) / Ci : ~0ull;      // phases (K + j Ci) u, j <= kcap, stay in the binade
          if (kcap > 0) {
            const double u = pow2d(e - 53);
            const uint64_t left = nmax - n - 1;                 // samples behind sample n that may still be taken
            const uint64_t jmax = kcap < left ? kcap : left;
            auto in_bin = [&](const uint64_t j) { return (uint32_t)(((double)(K + j * Ci) * u) * double_nbin) == ibin; };
            uint64_t j = jmax;                                  // last j <= jmax whose sample is in the bin (j = 0 is)
            if (!in_bin(jmax)) {
              // the bin number is monotone in j: bracket the last sample of the bin, starting from where the boundary
              // (ibin + 1) / nbin should be crossed, then bisect what is left (usually nothing)
              uint64_t lo = 0, hi = jmax;                       // in_bin(lo), !in_bin(hi)
              const double est = (((double)ibin + 1.0) / double_nbin - phi) / ((double)Ci * u);
              if (est >= 2.0 && est < (double)jmax) {
                const uint64_t g = (uint64_t)est;
                if (in_bin(g)) { lo = g; if (g + 2 < hi && !in_bin(g + 2)) hi = g + 2; }
                else { hi = g; if (g >= 2 && in_bin(g - 2)) lo = g - 2; }
              }
              while (hi - lo > 1) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if (in_bin(mid)) lo = mid; else hi = mid;
              }
              j = lo;
            }
            n += j + 1;
            phi = j < kcap ? (double)(K + (j + 1) * Ci) * u : (double)(K + j * Ci) * u + pps;   // out of the binade: the rounded addition itself
            if (j < jmax || n == nmax) { *phi_io = phi; return n; }      // the next sample is in another bin, or none is left
            stepped = true;
          }
        }
      }
    }
    if (!stepped) {
      n++;
      phi += pps;
      if (n == nmax) { *phi_io = phi; return n; }
    }
    // the state in front of the next sample: is it still in the bin?
    if (!(phi >= 0.0 && phi < 1.0)) phi -= floor(phi);
    if ((uint32_t)(phi * double_nbin) != ibin) { *phi_io = phi; return n; }
  }
}
}  // namespace dspsr_amd

// the run list of the plan (test hook of fold_plan_run; dspsr_amd_fold_set_bins builds its plan with it): runs[i] = first sample,
// bin, samples; hits[nbin] +=
extern "C" int dspsr_amd_fold_binplan_runs(double phi, double phase_per_sample, uint32_t nbin, uint64_t ndat, uint64_t* run_offset,
                                           uint32_t* run_bin, uint64_t* run_hits, uint64_t cap, uint64_t* nruns, uint32_t* hits)
{
  if (!nbin || !nruns) return DSPSR_AMD_EINVAL;
  uint64_t nr = 0, idat = 0;
  uint32_t cur = nbin;
  while (idat < ndat) {
    uint32_t ibin;
    const uint64_t n = dspsr_amd::fold_plan_run(&phi, phase_per_sample, double(nbin), ndat - idat, &ibin);
    if (ibin >= nbin) return DSPSR_AMD_EINVAL;
    if (ibin != cur) {
      if (nr < cap) { run_offset[nr] = idat; run_bin[nr] = ibin; run_hits[nr] = 0; }
      nr++;
      cur = ibin;
    }
    if (nr - 1 < cap) run_hits[nr - 1] += n;
    if (hits) hits[ibin] += (uint32_t)n;
    idat += n;
  }
  *nruns = nr;
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_fold_binplan(double phi, double phase_per_sample, uint32_t nbin, uint64_t ndat,
                                      uint32_t* binplan, uint32_t* hits)         // Fold.C:744-787
{
  if (!nbin || (!binplan && ndat)) return DSPSR_AMD_EINVAL;
  const double double_nbin = double(nbin);
  for (uint64_t idat = 0; idat < ndat; idat++) {
    phi -= floor(phi);
    const unsigned ibin = unsigned(phi * double_nbin);
    phi += phase_per_sample;
    if (ibin >= nbin) return DSPSR_AMD_EINVAL;
    binplan[idat] = ibin;
    if (hits) hits[ibin]++;
  }
  return DSPSR_AMD_OK;
}
