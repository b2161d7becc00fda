/* dspsr_amd.h -- C-ABI of the MI355X (gfx950) coherent-dedispersion + detection + fold engine.
 *
 * This is the drop-in boundary for DSPSR's GPU hot path.  Each entry point replaces one method of
 * the reference's Engine plug-in interfaces (citations are relative to the demorest/dspsr tree):
 *
 *   dsp::Memory                 Kernel/Classes/dsp/Memory.h:18-34      (CUDA impl MemoryCUDA.C:47-106)
 *   dsp::Filterbank::Engine     Signal/General/dsp/FilterbankEngine.h:15-44 (CUDA impl FilterbankCUDA.cu:73-304)
 *   dsp::Detection::Engine      Signal/General/dsp/Detection.h:98-106  (CUDA impl DetectionCUDA.cu:127-322)
 *   dsp::Fold::Engine           Signal/Pulsar/dsp/Fold.h:249-312       (CUDA impl FoldCUDA.cu:64-697)
 *   host-side preparation       Dedispersion.C:216-556, Response.C:132-344,649-700, optimize_fft.c:63-127,
 *                               Filterbank.C:55-263, Fold.C:650-787
 *
 * Conventions: plain C, opaque handles, every function returns 0 on success or a negative
 * DSPSR_AMD_E* code (never throws); dspsr_amd_last_error() gives the text a host wrapper turns
 * into the reference's `Error` exception.  All device work is enqueued on the context's HIP
 * stream and is asynchronous unless stated; pointers named *_dev are device pointers.
 * The host adaptor classes that bind these to the dsp::*::Engine interfaces are in
 * dspsr_amd/host/ (see INTEGRATION.md).
 */
#ifndef DSPSR_AMD_H
#define DSPSR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSPSR_AMD_OK 0
#define DSPSR_AMD_EINVAL (-1)   /* invalid argument / unsupported configuration (reference: Error InvalidParam/InvalidState) */
#define DSPSR_AMD_EHIP (-2)     /* HIP runtime failure (reference: CUFFTError / check_error) */
#define DSPSR_AMD_ENOMEM (-3)
#define DSPSR_AMD_ESTATE (-4)   /* called out of order (e.g. perform before set_kernel) */

typedef struct dspsr_amd_ctx dspsr_amd_ctx;
typedef struct dspsr_amd_filterbank dspsr_amd_filterbank;
typedef struct dspsr_amd_fold dspsr_amd_fold;

/* ---- context: one per pipeline thread / GPU, bound to one stream (SingleThread.C:213-290) ---- */
/* hip_stream: a hipStream_t to enqueue on (NULL = the legacy default stream), or
 * DSPSR_AMD_NEW_STREAM to let the context create and own a non-blocking stream */
#define DSPSR_AMD_NEW_STREAM ((void*)(intptr_t)-1)
int dspsr_amd_ctx_create(int device, void* hip_stream, dspsr_amd_ctx** ctx);
void dspsr_amd_ctx_destroy(dspsr_amd_ctx* ctx);
const char* dspsr_amd_last_error(const dspsr_amd_ctx* ctx);
int dspsr_amd_stream_sync(dspsr_amd_ctx* ctx);            /* FilterbankEngine::finish / check_error_stream */
const char* dspsr_amd_version(void);
/* sha256 (12 hex digits) of the sources the loaded library was built from (dspsr_amd/csrc/Makefile: BUILD_ID); the evidence files
 * under profiles/ and the bench line name it, so that a counter profile can be matched to the library that was timed */
const char* dspsr_amd_build_id(void);

/* ---- dsp::Memory (Memory.h:18-34): do_allocate / do_free / do_zero / do_copy ---- */
int dspsr_amd_malloc(dspsr_amd_ctx* ctx, size_t nbytes, void** ptr_dev);
int dspsr_amd_free(dspsr_amd_ctx* ctx, void* ptr_dev);
int dspsr_amd_zero(dspsr_amd_ctx* ctx, void* ptr_dev, size_t nbytes);
#define DSPSR_AMD_H2D 1
#define DSPSR_AMD_D2H 2
#define DSPSR_AMD_D2D 3
int dspsr_amd_copy(dspsr_amd_ctx* ctx, void* dst, const void* src, size_t nbytes, int kind);

/* ---- dsp::TimeSeries::Engine::copy_data_fpt (Kernel/Classes/dsp/TimeSeries.h:211-223; CUDA twin
 * Kernel/Classes/TimeSeriesCUDA.cu:20-29,75-200): device-to-device copy of `nfloat` floats of every
 * (channel, polarisation) row -- the overlap carry-over of dsp::InputBuffering (InputBuffering.C:35-126) and
 * TimeSeries::prepend.  Strides in floats between channel rows / polarisation rows; the row pointers already
 * include the start sample.  Rows of `to` and `from` must not overlap. */
int dspsr_amd_copy_fpt(dspsr_amd_ctx* ctx, float* to_dev, uint64_t to_chan_stride, uint64_t to_pol_stride,
                       const float* from_dev, uint64_t from_chan_stride, uint64_t from_pol_stride,
                       uint32_t nchan, uint32_t npol, uint64_t nfloat);
/* dsp::TimeSeries::operator += on device rows (Kernel/Classes/TimeSeries.C, used by PhaseSeries::combine,
 * Signal/Pulsar/PhaseSeries.C:442-484): to[row][i] += from[row][i]; same row geometry arguments as copy_fpt.  The profile
 * half of combining two device-resident PhaseSeries (the hits / integration_length half stays with the host object). */
int dspsr_amd_add_fpt(dspsr_amd_ctx* ctx, float* to_dev, uint64_t to_chan_stride, uint64_t to_pol_stride,
                      const float* from_dev, uint64_t from_chan_stride, uint64_t from_pol_stride,
                      uint32_t nchan, uint32_t npol, uint64_t nfloat);

/* ---- dsp::Filterbank::Engine ------------------------------------------------------------
 * setup(Filterbank*)  -> dspsr_amd_filterbank_create + dspsr_amd_filterbank_set_kernel
 *   (FilterbankCUDA.cu:73-168 reads freq_res, nchan_subband, input state, response nchan/ndat/
 *    impulse_pos/neg and copies the host-built, already swapped kernel to the device) */
typedef struct {
  uint32_t nchan_subband;   /* output channels per input channel          Filterbank.C:68
                               2^k, or 2^k times an odd R <= 127 (dspsr -F 96:D, -F 400:D, -F 1000:D, -F 25:D: the reference plans any
                               length, Filterbank.C:107-155; here the forward transform of R interleaved power-of-two
                               sub-sequences + one radix-R step -- kernels of their own for R = 3, 5, 7, 9, 15, a run-time-radix
                               form for the others; freq_res <= 8192 then) */
  uint32_t freq_res;        /* response ndat = backward FFT length                            Filterbank.C:93
                               1: the NON-CONVOLVING filterbank of `dspsr -F N` (Filterbank::Config::After / Never; Filterbank.C:614-623,
                               FilterbankCUDA.cu:92-116 with plan_bwd == NULL): one nchan_subband-point forward transform per output
                               sample, no backward transform, nkeep = 1, no overlap (nfilt_pos = nfilt_neg = 0); nchan_subband 2^k in
                               [2, 8192]; a kernel of input_nchan * nchan_subband factors is applied if set (Response::operate); one
                               launch, no scratch (csrc/fb_plain.hip).  dsp::Convolution then runs on its output as a second object
                               with nchan_subband = 1.
                               2^k >= 2, or 2^k times an odd R <= 127 (dspsr -x 12288, -x 11264): bins R m' + r of a channel are R
                               pseudo-channels of freq_res / R bins through the power-of-two passes, one radix-R step in time adds
                               their transforms (a pass of its own: fold_is_fused() == 0; freq_res / R <= 8192).  Both lengths may
                               carry an odd factor when the product of the two stays <= 127 (-F 96:D -x 768) */
  uint32_t nfilt_pos;       /* response impulse_pos                       Filterbank.C:90  */
  uint32_t nfilt_neg;       /* response impulse_neg                       Filterbank.C:91  */
  uint32_t input_nchan;     /* input channels (kernel has input_nchan*nchan_subband*freq_res bins) */
  uint32_t npol;            /* 1 or 2 */
  uint32_t real_input;      /* 1: Signal::Nyquist (ndim 1), 0: Signal::Analytic (ndim 2) */
  uint32_t max_parts;       /* parts processed per launch group (scratch is sized for this); 0 => default */
  uint32_t force_four_pass; /* 0 => passes chosen from the geometry; 1: two-pass inverse (the path of freq_res > 8192 and of
                               dsp::Convolution) also where the single-pass inverse would do -- and instead of the one-pass /
                               three-pass convolution; 2: never the two-pass path of short responses (complex dual-pol input,
                               nchan_subband * freq_res^2 <= 2^27: forward and inverse transforms in two tiles), the one-pass /
                               three-pass convolution or the grouping of a convolution's channels; same results to rounding in every case */
  uint32_t fused_fold;      /* dspsr_amd_filterbank_perform_fold: DSPSR_AMD_FUSED_AUTO (fold inside the last filterbank
                               pass when the channel tiles fill the chip), _ALWAYS, _NEVER -- same sums bit for bit */
} dspsr_amd_filterbank_config;
#define DSPSR_AMD_FUSED_AUTO 0
#define DSPSR_AMD_FUSED_ALWAYS 1
#define DSPSR_AMD_FUSED_NEVER 2

int dspsr_amd_filterbank_create(dspsr_amd_ctx* ctx, const dspsr_amd_filterbank_config* cfg,
                                dspsr_amd_filterbank** fb);
void dspsr_amd_filterbank_destroy(dspsr_amd_filterbank* fb);
/* host kernel: input_nchan*nchan_subband*freq_res complex floats (response->get_datptr(0,0)); NULL => no response */
int dspsr_amd_filterbank_set_kernel(dspsr_amd_filterbank* fb, const float* kernel_host, uint64_t ncomplex);
/* derived sizes, same arithmetic as Filterbank::make_preparations (Filterbank.C:107-155) */
int dspsr_amd_filterbank_sizes(const dspsr_amd_filterbank* fb, uint64_t* nsamp_fft, uint64_t* nsamp_overlap,
                               uint64_t* nsamp_step, uint32_t* nkeep);

/* perform(in, out, npart, in_step, out_step)  (FilterbankEngine.h:28-32, FilterbankCUDA.cu:181-304)
 *   in_dev : unpacked float32 rows, in->get_datptr(ichan,ipol) = in_dev + ichan*in_chan_stride + ipol*in_pol_stride (floats)
 *   out_dev: complex float rows, out->get_datptr(ochan,ipol) = out_dev + ochan*out_chan_stride + ipol*out_pol_stride (floats);
 *            NULL => benchmark only (FilterbankCUDA.cu:265)
 *   in_step: floats between parts (= nsamp_step*ndim), out_step: floats between parts (= 2*nkeep) */
int dspsr_amd_filterbank_perform(dspsr_amd_filterbank* fb, const float* in_dev, uint64_t in_chan_stride,
                                 uint64_t in_pol_stride, float* out_dev, uint64_t out_chan_stride,
                                 uint64_t out_pol_stride, uint64_t npart, uint64_t in_step, uint64_t out_step);

/* Optional side channel (north star: fuse the 8-bit load into FFT pass 1; replaces the separate
 * unpack kernels GenericEightBitUnpackerCUDA.cu:24-45 / CASPSRUnpackerCUDA.cu:44-82 + perform).
 * raw_dev points at the first byte of the block (BitSeries::get_rawptr()); value = (int8+0.5)*scale. */
#define DSPSR_AMD_RAW_GENERIC 0  /* byte ((t*nchan+c)*npol+p)*ndim+d   BitUnpacker.C:48-80 */
#define DSPSR_AMD_RAW_CASPSR 1   /* 4 B pol0, 4 B pol1 repeating       CASPSRUnpacker.C:132-187; the block starts on a group
                                    boundary and holds whole 8-byte groups: ceil(nsamp/4)*8 bytes */
#define DSPSR_AMD_RAW_UWB16 2    /* 16-bit offset-binary complex, 2048-sample blocks per polarisation, single channel;
                                    value = float(int16(x ^ 0x8000)) * scale   uwb/UWBUnpackerCUDA.cu:24-75,
                                    uwb/UWBUnpacker.C:196-207 (the reference applies no scale: pass 1.0) */
int dspsr_amd_filterbank_perform_raw(dspsr_amd_filterbank* fb, const int8_t* raw_dev, int raw_layout, float scale,
                                     float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride,
                                     uint64_t npart, uint64_t out_step);

/* Fused filterbank + Detection::polarimetry (npol must be 2): writes the detected TimeSeries directly.
 *   state: DSPSR_AMD_COHERENCE | DSPSR_AMD_STOKES ; ndim in {1,2,4} (Detection.C:423-474 layouts):
 *     ndim 4: det_dev + chan*det_chan_stride + 4*idat               (npol 1)
 *     ndim 2: det_dev + chan*det_chan_stride + plane*det_pol_stride + 2*idat   (plane0 = PP,QQ plane1 = Re,Im)
 *     ndim 1: det_dev + chan*det_chan_stride + k*det_pol_stride + idat         (k = 0..3)
 *   exactly one of in_f32_dev / raw_dev is non-NULL. */
#define DSPSR_AMD_COHERENCE 0
#define DSPSR_AMD_STOKES 1
int dspsr_amd_filterbank_perform_detect(dspsr_amd_filterbank* fb, const float* in_f32_dev, uint64_t in_chan_stride,
                                        uint64_t in_pol_stride, uint64_t in_step, const int8_t* raw_dev,
                                        int raw_layout, float scale, int state, uint32_t ndim, float* det_dev,
                                        uint64_t det_chan_stride, uint64_t det_pol_stride, uint64_t npart);

/* Fused Filterbank -> Detection -> Fold: the detected time series never leaves the chip.  The profile of `fold` is
 * either npol 1 x ndim 4 (one (PP, QQ, Re, Im) float4 per bin: Detection's CPU default, LoadToFoldConfig.C:104) or
 * npol 2 x ndim 2 (rows (PP, QQ) and (Re, Im) per channel: what the reference's GPU pipeline detects and folds,
 * LoadToFold1.C:1105-1109, DetectionCUDA.cu:145-149) -- the same sums in either shape.  Replaces the chain Filterbank::Engine::perform (FilterbankCUDA.cu:181-304) + Detection::Engine::polarimetry
 * (DetectionCUDA.cu:127-177) + Fold::Engine::fold (FoldCUDA.cu:586-697) for one block of `npart` parts.
 * The bin plan of the npart*nkeep output samples of this call must have been handed to `fold` beforehand
 * (dspsr_amd_fold_set_nbin / set_ndat / set_bin(s) with sample indices counted from the first output sample of
 * this call); it is consumed.  Sums are accumulated into the device profile of `fold` in time order per
 * (chan, bin), bit-identical to perform_detect followed by dspsr_amd_fold_fold.
 * dspsr_amd_filterbank_fold_is_fused() says how this object folds (fused_fold = DSPSR_AMD_FUSED_AUTO):
 *   1  inside the last filterbank pass, one workgroup per tile of channels walking the parts in order (three-pass geometry
 *      with at least one tile per compute unit): exact time-order sums, bit-identical to Detection + Fold;
 *   2  inside the last pass with the parts of a launch cut into runs folded by different workgroups (8 .. ncu-1 tiles):
 *      run 0 continues the profile, the other runs go to partial profiles added in run order after the launch -- sums
 *      re-associated per run, equal to the time-order sums to float rounding, deterministic;
 *   3  four-pass geometry (freq_res > 8192, dsp::Convolution shapes) and phase bins of >= 64 samples: the second inverse pass
 *      reduces every run of 32 consecutive output samples it holds to (at most two) piece sums, a second kernel adds the
 *      pieces of each (channel, bin) in time order -- re-associated like the long-run fold (fold.hip), deterministic; the
 *      detected time series never reaches HBM.  Plans that do not qualify (narrow bins, zero weights, partial folds) take 0;
 *   0  Detection and Fold as separate launches on a block owned by the filterbank object (fewer than 8 tiles,
 *      DSPSR_AMD_FUSED_NEVER, or what the other modes turn down; and every dsp::Convolution object whose float32 rows run in one
 *      or three tile passes -- complex input, two polarisations, 64 <= freq_res <= 2^21: Detection in their last pass).  Three-pass plans with runs of >= 640 samples per bin
 *      take this path too (fold.hip).
 * DSPSR_AMD_FUSED_ALWAYS forces mode 1 on any three-pass geometry. */
int dspsr_amd_filterbank_fold_is_fused(const dspsr_amd_filterbank* fb);
/* How many transform passes (trips of the part through HBM scratch + 1) a call makes -- the role of the plan choice inside
 * CUDA::FilterbankEngine::setup (FilterbankCUDA.cu:92-116: one forward and one batched backward cuFFT plan).  raw_input != 0:
 * the answer for dspsr_amd_filterbank_perform_raw / _detect / _fold on a generic 8-bit block, else for float32 rows.
 *   1  freq_res = 1: the non-convolving filterbank, one tile pass from the input to the output rows; and (raw_input == 0)
 *      dsp::Convolution shapes -- nchan_subband = 1, complex input with two polarisations -- with 64 <= freq_res <= 8192 on float32
 *      rows: forward transform, response, backward transform, keep window and Detection of a (channel, part) sequence in ONE tile
 *      (csrc/fb_conv1.hip; force_four_pass != 0 keeps the four passes);
 *   2  short responses: complex dual-pol 8-bit input with 512 <= freq_res <= 4096 and 2^13 / freq_res <= nchan_subband <=
 *      2^27 / freq_res^2 (upper end: one 50 MHz sub-band with -F 512:D -x 512): column forward pass, then rows + chirp + inverse
 *      transforms in ONE tile -- the spectrum stays on chip;
 *   3  forward columns, forward rows, inverse per channel (freq_res <= 8192); and (raw_input == 0) the dsp::Convolution shapes above
 *      with 2^14 <= freq_res <= 2^21 on float32 rows: the forward transform's second pass and the inverse transform's first one run
 *      along the same rows of the spectrum and are ONE pass, the spectrum never reaches memory (csrc/fb_conv3.hip;
 *      force_four_pass != 0 keeps the four passes);
 *   4  two-pass inverse (freq_res > 8192, dsp::Convolution shapes, force_four_pass = 1).  dsp::Convolution on >= 4 complex channels
 *      of float32 rows runs GROUPS of channels as one launch group (forward passes per channel, inverse passes of a group-wide
 *      filterbank: the same numbers bit for bit; force_four_pass = 2 keeps the loop over the channels).
 * The count is that of the TILE passes.  Lengths with an odd factor run the three tile passes per power-of-two sub-sequence and
 * add trips through HBM outside the tiles: nchan_subband = R * 2^k a de-interleave of the input (k_sub_split) and one radix-R pass
 * over the spectrum (k_sub_combine); freq_res = R * 2^k a radix-R pass over the pseudo-channels' time series as well
 * (k_time_combine, with Detection and Fold as launches of their own: fold_is_fused() == 0). */
int dspsr_amd_filterbank_npass(const dspsr_amd_filterbank* fb, int raw_input);
int dspsr_amd_filterbank_perform_fold(dspsr_amd_filterbank* fb, const float* in_f32_dev, uint64_t in_chan_stride,
                                      uint64_t in_pol_stride, uint64_t in_step, const int8_t* raw_dev, int raw_layout,
                                      float scale, int state, dspsr_amd_fold* fold, uint64_t npart);

/* ---- digifil with a convolving filterbank, `digifil -F N:D [-x M] -t T` (Signal/General/LoadToFil.C:185-222,250-304) ---------
 * One launch group for dsp::Filterbank (+ Dedispersion response) -> dsp::Detection::square_law (Detection.C:218-320: Intensity
 * = (Re0^2 + Im0^2) + (Re1^2 + Im1^2), or PPQQ) -> dsp::TScrunch, FPT branch (TScrunch.C:128-178): the detected rows are never
 * written at the filterbank's output rate.  The npart * nkeep detected samples of every channel continue a STREAM: TScrunch's
 * input buffering re-presents the ndat % tscrunch left-over samples in front of the next block (TScrunch.C:110-111); here the
 * partial sum of those samples stays in carry_dev -- the same sequential sum, out = in[0]; out += in[1]; ... , bit for bit.
 *   out_state       DSPSR_AMD_INTENSITY (npol_out 1) | DSPSR_AMD_PPQQ (npol_out 2)
 *   out_dev         rows [chan][npol_out] of *nout floats, FPT order like the reference's TimeSeries
 *   carry_dev       [nchan][npol_out] floats owned by the caller (any content when *carry_count == 0)
 *   carry_count     in: samples already summed into carry_dev (< tscrunch); out: samples left over by this call
 *   nout            out: complete output samples written per row = (carry_in + npart * nkeep) / tscrunch
 * tscrunch = 1 writes the detected samples themselves.  Three-pass and two-pass geometries run it inside the inverse pass; the
 * others (freq_res > 8192, odd factors) through an internal detected block + dspsr_amd_tscrunch_fpt -- the same numbers. */
#define DSPSR_AMD_INTENSITY 2
#define DSPSR_AMD_PPQQ 3
int dspsr_amd_filterbank_perform_search(dspsr_amd_filterbank* fb, const float* in_f32_dev, uint64_t in_chan_stride,
                                        uint64_t in_pol_stride, uint64_t in_step, const int8_t* raw_dev, int raw_layout,
                                        float scale, int out_state, uint32_t tscrunch, float* out_dev, uint64_t out_chan_stride,
                                        uint64_t out_pol_stride, float* carry_dev, uint32_t* carry_count, uint64_t npart,
                                        uint64_t* nout);
/* 1: dspsr_amd_filterbank_perform_search runs inside the inverse pass for this object; 0: through the internal detected block */
int dspsr_amd_filterbank_search_is_fused(const dspsr_amd_filterbank* fb);
/* dsp::TScrunch::fpt_tscrunch (TScrunch.C:148-178; TScrunch::Engine::fpt_tscrunch, dsp/TScrunch.h:61-69, CUDA twin TScrunchCUDA.cu:
 * 204-300: ndim 1 or 2) on device rows [nchan][npol] of ndat_in samples of ndim floats each, every dimension on its own, as a stream:
 * carry / carry_count / nout as above.  Out of place only (digifil scrunches in place on the host, LoadToFil.C:296-304: here every
 * output sample has its own thread). */
int dspsr_amd_tscrunch_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                           float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan, uint32_t npol,
                           uint32_t ndim, uint64_t ndat_in, uint32_t sfactor, float* carry_dev /* [nchan][npol][ndim] */,
                           uint32_t* carry_count, uint64_t* nout);
/* dsp::FScrunch::fpt_fscrunch (FScrunch.C:117-145; FScrunch::Engine::fpt_fscrunch, dsp/FScrunch.h:56-64, CUDA twin FScrunchCUDA.cu:
 * 50-90): out row (c, p) = in row (c*sfactor, p); += rows c*sfactor + 1 ... in order.
 * nchan_in must be a multiple of sfactor; out of place only. */
int dspsr_amd_fscrunch_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                           float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan_in, uint32_t npol,
                           uint64_t nfloat, uint32_t sfactor);

/* ---- dsp::Detection::Engine (Detection.h:98-106) ------------------------------------------
 * polarimetry(ndim, in, out): in = complex rows [nchan][2][ndat]; in-place allowed for ndim 2
 * (LoadToFold1.C:545-546 uses input==output).  Layouts as above. */
int dspsr_amd_detect_polarimetry(dspsr_amd_ctx* ctx, int state, uint32_t ndim, const float* in_dev,
                                 uint64_t in_chan_stride, uint64_t in_pol_stride, float* out_dev,
                                 uint64_t out_chan_stride, uint64_t out_pol_stride, uint32_t nchan, uint64_t ndat);
/* square_law(in,out) (Detection.C:218-320): out[chan][pol][idat] = re^2+im^2 ; intensity!=0 sums the two pols */
int dspsr_amd_detect_square_law(dspsr_amd_ctx* ctx, int intensity, const float* in_dev, uint64_t in_chan_stride,
                                uint64_t in_pol_stride, float* out_dev, uint64_t out_chan_stride,
                                uint64_t out_pol_stride, uint32_t nchan, uint32_t npol, uint64_t ndat);

/* ---- search mode (digifil): dsp::TFPFilterbank (TFPFilterbank.C:27-101: forward FFT of 2*nchan real
 * samples per pol and part, Re^2+Im^2 of bins 0..nchan-1, TFP order) + optional pol sum + dsp::TScrunch
 * (TScrunch.C:180-206) fused in one launch.  Real dual-pol 8-bit input, raw_dev = first byte of the block.
 *   out_dev: [npart/tscrunch][nchan][pscrunch ? 1 : 2] floats (PPQQ or Intensity, TimeSeries::OrderTFP) */
typedef struct {
  uint32_t nchan;      /* -F nchan (power of two, 16..8192: both polarisations of one part fill a 2^14-point tile at 8192) */
  uint32_t npol;       /* input polarisations (2) */
  uint32_t pscrunch;   /* 1: Intensity (p0+p1), 0: PPQQ */
  uint32_t tscrunch;   /* -t factor, 0/1 = none */
} dspsr_amd_tfp_config;
int dspsr_amd_tfp_filterbank(dspsr_amd_ctx* ctx, const dspsr_amd_tfp_config* cfg, const int8_t* raw_dev, int raw_layout,
                             float scale, float* out_dev, uint64_t npart);

/* ---- dsp::Fold::Engine (Fold.h:249-312, FoldCUDA.cu) --------------------------------------
 * The engine owns the device-resident profile (get_profiles()).  Call order per Fold::fold
 * (Fold.C:724-829): set_nbin, set_ndat, set_bin x ndat (or set_bins), fold.  synch copies to host. */
int dspsr_amd_fold_create(dspsr_amd_ctx* ctx, dspsr_amd_fold** fold);
void dspsr_amd_fold_destroy(dspsr_amd_fold* fold);
/* shape of the input TimeSeries / output PhaseSeries (Fold::Engine::setup, Fold.C:973-1007) */
int dspsr_amd_fold_set_shape(dspsr_amd_fold* fold, uint32_t nchan, uint32_t npol, uint32_t ndim, uint32_t nbin);
/* Fold::Engine::setup (Fold.C:968-1011): fold INTO the engine-owned device PhaseSeries -- profile_dev =
 * get_profiles()->get_datptr(0,0), span_floats = get_nfloat_span() (floats between consecutive (chan, pol) rows, each
 * row nbin*ndim floats) -- so that Fold::prepare_output / zero / mixable (Fold.C:88-94,123-148,495-508), which act on
 * get_profiles(), see the sums.  The buffer stays the caller's; profile_dev = NULL returns to a library-owned profile. */
int dspsr_amd_fold_bind_profile(dspsr_amd_fold* fold, float* profile_dev, uint64_t span_floats, uint32_t nchan,
                                uint32_t npol, uint32_t ndim, uint32_t nbin);
int dspsr_amd_fold_set_nbin(dspsr_amd_fold* fold, uint32_t nbin);                        /* FoldCUDA.cu:64-70 */
int dspsr_amd_fold_set_ndat(dspsr_amd_fold* fold, uint64_t ndat, uint64_t idat_start);   /* FoldCUDA.cu:72-82 */
int dspsr_amd_fold_set_bin(dspsr_amd_fold* fold, uint64_t idat, double ibin, double bins_per_sample); /* :84-113 */
/* whole plan at once: the double recurrence of Fold.C:744-787 run inside the library;
 * hits_host[nbin] (may be NULL) is incremented like Fold.C:783; returns ndat folded via *ndat_folded */
int dspsr_amd_fold_set_bins(dspsr_amd_fold* fold, double phi, double phase_per_sample, uint64_t ndat,
                            uint64_t idat_start, uint32_t* hits_host, uint64_t* ndat_folded);
/* the same with the weights of the input (Fold.C:686-716,746-763: WeightedTimeSeries, one weight per ndatperweight samples,
 * sample idat belongs to weight (idat + weight_idat) / ndatperweight): samples of a zero weight are left out of the plan,
 * of hits[] and of *ndat_folded.  The hook for flagged / dropped data; spectral kurtosis itself is not part of this path. */
int dspsr_amd_fold_set_bins_weighted(dspsr_amd_fold* fold, double phi, double phase_per_sample, uint64_t ndat,
                                     uint64_t idat_start, const uint32_t* weights_host, uint64_t nweights,
                                     uint64_t ndatperweight, uint64_t weight_idat, uint32_t* hits_host,
                                     uint64_t* ndat_folded);
/* fold(): accumulate in_dev rows (get_datptr(ichan,ipol) = in_dev + ichan*in_chan_stride + ipol*in_pol_stride)
 * into the device profile using the plan built since the last set_nbin (FoldCUDA.cu:586-697) */
int dspsr_amd_fold_fold(dspsr_amd_fold* fold, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride);
/* fold() of an input that carries zeroed (RFI-excised) samples -- Fold::Engine::zeroed_samples with hits_nchan == nchan
 * (Fold.C:853-866, fold1bin*hits FoldCUDA.cu:415-576,622): as dspsr_amd_fold_fold, and hits_dev[ichan*nbin + ibin] (device,
 * uint32) is incremented by the number of planned samples of polarisation 0 whose first float is not zero.  The hook the
 * reference's spectral-kurtosis chain needs; the excision itself is not part of this path. */
int dspsr_amd_fold_fold_zeroed(dspsr_amd_fold* fold, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                               uint32_t* hits_dev);
float* dspsr_amd_fold_profiles_dev(dspsr_amd_fold* fold);   /* device [nchan][npol][nbin][ndim] (get_profiles) */
uint64_t dspsr_amd_fold_get_ndat_folded(const dspsr_amd_fold* fold);
int dspsr_amd_fold_zero(dspsr_amd_fold* fold);                                            /* Engine::zero */
int dspsr_amd_fold_synch(dspsr_amd_fold* fold, float* profile_host);                      /* FoldCUDA.cu:127-152 (blocks) */

/* ---- the sub-integration dump over RCCL / xGMI: the ONE exchange of the path -----------------------------------------
 * Reference hook: dsp::Subint<Fold>::transformation emits the finished sub-integration (Signal/Pulsar/dsp/Subint.h:291-303);
 * merging semantics = PhaseSeries::combine (Signal/Pulsar/PhaseSeries.C:442-484); the reference merges its threads' pieces
 * on the host today (Signal/General/MultiThread.C:274-379).  One process per GPU, one communicator per pipeline context.
 *   dspsr_amd_comm_unique_id : ncclGetUniqueId -- rank 0 calls it and hands the 128 bytes to every rank by any host means
 *                              (MPI, a file, a socket; dspsr's own MPI transport, Kernel/Classes/mpi)
 *   dspsr_amd_comm_create    : ncclCommInitRank on the context's device (collective: every rank calls it)
 *   dspsr_amd_reduce_profiles_start : snapshot of this rank's device profile rows (`nrow` rows of `row_floats` floats,
 *       `span_floats` apart -- Fold::Engine::get_profiles()) on the context's stream, then ONE collective on the
 *       communicator's own stream, so the caller may zero the profile and launch the next block at once:
 *         DSPSR_AMD_REDUCE_SUM    time-slice replicas: profile, hits[], ndat_total and integration_length all SUMMED onto
 *                                 `root` in one packed ncclReduce
 *         DSPSR_AMD_REDUCE_GATHER sub-band shards (rank g = input channel g): the ranks' slices delivered to `root` in rank
 *                                 order (ncclGather); hits[] / lengths are identical on every rank and are the root's own.
 *                                 check_hits != 0 adds a MIN/MAX all-reduce of hits[] to the same group.
 *   dspsr_amd_reduce_profiles_finish : waits for the exchange.  On `root`: profile_host receives nrow*row_floats floats
 *       (SUM) or nranks*nrow*row_floats floats (GATHER), packed; hits_host[nbin], *integration_length, *ndat_total the
 *       merged values.  Other ranks receive nothing.  *hits_identical (every rank; may be NULL) = 0 if check_hits found
 *       ranks that disagree. */
typedef struct dspsr_amd_comm dspsr_amd_comm;
#define DSPSR_AMD_UNIQUE_ID_BYTES 128
#define DSPSR_AMD_REDUCE_SUM 0
#define DSPSR_AMD_REDUCE_GATHER 1
/* optional, before the first communicator: the RCCL shared object to open (default: librccl.so.1 from the loader path,
 * then /opt/rocm/lib).  A host process that carries its own ROCm runtime (PyTorch) names the RCCL built against it. */
int dspsr_amd_comm_set_library(const char* path);
int dspsr_amd_comm_unique_id(void* id_out);
int dspsr_amd_comm_create(dspsr_amd_ctx* ctx, int nranks, int rank, const void* unique_id, dspsr_amd_comm** comm);
void dspsr_amd_comm_destroy(dspsr_amd_comm* comm);
int dspsr_amd_comm_rank(const dspsr_amd_comm* comm);
int dspsr_amd_comm_size(const dspsr_amd_comm* comm);
int dspsr_amd_reduce_profiles_start(dspsr_amd_comm* comm, int mode, int root, const float* profile_dev, uint64_t span_floats,
                                    uint64_t nrow, uint64_t row_floats, const uint32_t* hits_host, uint32_t nbin,
                                    double integration_length, uint64_t ndat_total, int check_hits);
int dspsr_amd_reduce_profiles_finish(dspsr_amd_comm* comm, float* profile_host, uint32_t* hits_host, double* integration_length,
                                     uint64_t* ndat_total, int* hits_identical);
/* root, after finish: the merged profile where the exchange left it -- the communicator's pinned host buffer (*nfloat floats,
 * valid until the next start) -- for a writer that wants to read it in place (pass profile_host = NULL to finish) */
const float* dspsr_amd_reduce_profiles_result(const dspsr_amd_comm* comm, uint64_t* nfloat);

/* ---- integer-sample inter-channel delay (-K): dsp::SampleDelay (Signal/General/SampleDelay.C:52-195) --------------
 * create    : SampleDelay::build (:52-102) from the delay of each row, delays_host[ichan*npol+ipol] (the values
 *             SampleDelayFunction::get_delay returns); absolute = function->get_absolute()
 * transform : SampleDelay::transformation (:123-195): out row i = in row shifted by its applied delay, ndat_out =
 *             ndat_in - total_delay (0 if negative); in == out allowed (LoadToFold1.C:617-618); strides in floats.
 *             The caller shifts the start time by zero_delay samples (:159) and re-presents the last total_delay
 *             samples with the next block (InputBuffering, :117,146). */
typedef struct dspsr_amd_sample_delay dspsr_amd_sample_delay;
int dspsr_amd_sample_delay_create(dspsr_amd_ctx* ctx, uint32_t nchan, uint32_t npol, const int64_t* delays_host,
                                  int absolute, dspsr_amd_sample_delay** out);
void dspsr_amd_sample_delay_destroy(dspsr_amd_sample_delay* h);
int64_t dspsr_amd_sample_delay_zero_delay(const dspsr_amd_sample_delay* h);
uint64_t dspsr_amd_sample_delay_total_delay(const dspsr_amd_sample_delay* h);
int dspsr_amd_sample_delay_transform(dspsr_amd_sample_delay* h, const float* in_dev, uint64_t in_chan_stride,
                                     uint64_t in_pol_stride, float* out_dev, uint64_t out_chan_stride,
                                     uint64_t out_pol_stride, uint32_t ndim, uint64_t ndat_in, uint64_t* ndat_out);

/* ---- search-mode output stage: dsp::Rescale + dsp::SigProcDigitizer on TFP-ordered data ----------
 * dspsr_amd_rescale_*        : dsp::Rescale (Signal/General/Rescale.C:157-420), scalar offset/scale per (pol, chan)
 *                              re-estimated every `interval_samples` samples (0 = length of the first block) and once
 *                              right after the first call; `constant` keeps the first estimate (set_constant, :50).
 *                              in == out is allowed (digifil rescales in place, LoadToFil.C:325-326).
 * dspsr_amd_sigproc_digitize : dsp::SigProcDigitizer::pack, TFP branch (Kernel/Formats/sigproc/SigProcDigitizer.C:80-236)
 *                              nbit 1/2/4/8/16 or -32 (pack_float :309-342); flip_band = input bandwidth > 0,
 *                              swap_band = input->get_swap() (ChannelSort :38-66); input_scale = input->get_scale().
 * dspsr_amd_pscrunch_tfp     : dsp::PScrunch, TFP branch (Signal/General/PScrunch.C:36-90): (p0 + p1) * float(1/sqrt 2), the
 *                              step digifil runs between Rescale and the digitizer (LoadToFil.C:333-343); out of place. */
int dspsr_amd_pscrunch_tfp(dspsr_amd_ctx* ctx, const float* in_tfp_dev, float* out_tfp_dev, uint64_t ndat, uint32_t nchan,
                           uint32_t npol);
typedef struct dspsr_amd_rescale dspsr_amd_rescale;
int dspsr_amd_rescale_create(dspsr_amd_ctx* ctx, uint32_t nchan, uint32_t npol, uint64_t interval_samples, int constant,
                             dspsr_amd_rescale** out);
void dspsr_amd_rescale_destroy(dspsr_amd_rescale* r);
int dspsr_amd_rescale_transform(dspsr_amd_rescale* r, const float* in_tfp_dev, float* out_tfp_dev, uint64_t ndat);
int dspsr_amd_rescale_get(dspsr_amd_rescale* r, float* offset_host, float* scale_host);   /* [npol*nchan]: index ichan*npol+ipol */
/* digifil's whole output stage behind the TFP filterbank (LoadToFil.C:318-362) in one pass over a PPQQ block [ndat][nchan][2]:
 * dsp::Rescale (statistics and intervals exactly as dspsr_amd_rescale_transform, Rescale.C:217-388) -> dsp::PScrunch
 * (PScrunch.C:52,72-90) -> dsp::SigProcDigitizer::pack (SigProcDigitizer.C:112-236; nbit 1/2/4/8/16, input scale 1 behind Rescale),
 * the same float operations in the same order as the three separate calls -- identical bytes -- without writing the rescaled
 * and the summed block.  out_dev: [ndat][nchan * nbit / 8] bytes. */
int dspsr_amd_rescale_pscrunch_digitize(dspsr_amd_rescale* r, const float* in_tfp_dev, uint64_t ndat, int nbit, float scale_fac,
                                        int flip_band, int swap_band, void* out_dev);
int dspsr_amd_sigproc_digitize(dspsr_amd_ctx* ctx, const float* in_tfp_dev, uint64_t ndat, uint32_t nchan, uint32_t npol,
                               int nbit, int use_digi_scales, double input_scale, float scale_fac, int flip_band,
                               int swap_band, void* out_dev);
/* The same two operations on FPT-ordered rows [nchan][npol] of ndat floats -- what digifil's convolving branch hands them (the
 * Filterbank keeps the reference's FPT order): Rescale.C:232-262,330-347 (FPT branches; the statistics, intervals and offset /
 * scale arrays are those of the TFP form: one dspsr_amd_rescale object serves either order, index ichan*npol + ipol) and
 * SigProcDigitizer.C:238-290 (FPT branch: the bytes leave in the same TPF order as from TFP input, outidx = idat*nchan*npol +
 * ipol*nchan + ichan; pack_float :346-358 for nbit -32).
 * dspsr_amd_rescale_digitize_fpt: both in one pass over the rows (the rescaled block is not written), identical bytes. */
int dspsr_amd_rescale_transform_fpt(dspsr_amd_rescale* r, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                    float* out_dev, uint64_t out_chan_stride, uint64_t out_pol_stride, uint64_t ndat);
int dspsr_amd_sigproc_digitize_fpt(dspsr_amd_ctx* ctx, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                   uint64_t ndat, uint32_t nchan, uint32_t npol, int nbit, int use_digi_scales, double input_scale,
                                   float scale_fac, int flip_band, int swap_band, void* out_dev);
int dspsr_amd_rescale_digitize_fpt(dspsr_amd_rescale* r, const float* in_dev, uint64_t in_chan_stride, uint64_t in_pol_stride,
                                   uint64_t ndat, int nbit, float scale_fac, int flip_band, int swap_band, void* out_dev);

/* ---- host-side preparation (stays on the host in the reference as well) --------------------- */
typedef struct {
  double centre_frequency;   /* MHz */
  double bandwidth;          /* MHz, signed */
  double dispersion_measure;
  uint32_t input_nchan;
  uint32_t nchan;            /* output channels */
  uint32_t ndim;             /* input ndim: 1 Nyquist, 2 Analytic */
  int32_t dual_sideband;     /* -1 unset (Observation.C:80-87) */
  uint32_t dc_centred;
  uint32_t swap;
  uint32_t freq_res;         /* 0 => optimal (optimize_fft.c) else -x value (checked like Response.C:328-344) */
  uint32_t ndat_max;         /* Response::ndat_max, 0 => none */
  uint32_t fractional_delay; /* -K: add the fractional-sample inter-channel delay phase (Dedispersion.C:524-545,
                                set by LoadToFold1.C:605-624); the integer part is dsp::SampleDelay's job */
} dspsr_amd_dedispersion_config;

typedef struct {
  uint32_t impulse_pos, impulse_neg;   /* Dedispersion.C:216-248 */
  uint32_t minimum_ndat;               /* Response.C:259-275 */
  uint32_t ndat;                       /* chosen frequency resolution */
} dspsr_amd_dedispersion_info;

/* Dedispersion::prepare + resolution choice; error if -x is below the minimum (Response::check_ndat) */
int dspsr_amd_dedispersion_prepare(const dspsr_amd_dedispersion_config* cfg, dspsr_amd_dedispersion_info* info,
                                   char* errbuf, size_t errlen);
/* Dedispersion::build + Response::match ordering (Dedispersion.C:261-331,478-556; Response.C:132-181):
 * kernel_host receives nchan*ndat complex floats */
int dspsr_amd_dedispersion_build(const dspsr_amd_dedispersion_config* cfg, uint32_t ndat, float* kernel_host);
/* Dedispersion::SampleDelay::match (DedispersionSampleDelay.C:24-75): delay of each channel in samples relative to the
 * centre frequency, channel frequencies as Observation::get_centre_frequency(ichan) (Observation.C:420-451) */
int dspsr_amd_dedispersion_sample_delays(double centre_frequency, double bandwidth, double dispersion_measure,
                                         uint32_t nchan, double rate_hz, int swap, uint32_t nsub_swap, int dc_centred,
                                         int64_t* delays_host);
uint64_t dspsr_amd_optimal_fft_length(uint64_t nbadperfft, uint64_t nfft_max);           /* optimize_fft.c:63-127 */
/* (int8+0.5)*scale constant of the 8-bit LUT (BitTable.C:165-218) for a given JA98 spacing (ext) */
double dspsr_amd_eight_bit_scale(double ja98_spacing);
/* bin plan of Fold.C:744-787: binplan[ndat] and hits[nbin] += */
int dspsr_amd_fold_binplan(double phi, double phase_per_sample, uint32_t nbin, uint64_t ndat,
                           uint32_t* binplan_host, uint32_t* hits_host);
/* the same plan as its runs (first sample, bin, samples; at most `cap` stored, *nruns = how many there are), found per run
 * instead of per sample -- the values of the recurrence, not an approximation (csrc/host_prep.cpp fold_plan_run); what
 * dspsr_amd_fold_set_bins builds its plan with */
int dspsr_amd_fold_binplan_runs(double phi, double phase_per_sample, uint32_t nbin, uint64_t ndat, uint64_t* run_offset,
                                uint32_t* run_bin, uint64_t* run_hits, uint64_t cap, uint64_t* nruns, uint32_t* hits_host);

#ifdef __cplusplus
}
#endif
#endif /* DSPSR_AMD_H */
