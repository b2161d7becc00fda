// Internals of the fold engine shared with the fused filterbank+detect+fold path (not installed).
#pragma once
#include <vector>

#include "engine_internal.h"

namespace dspsr_amd {
struct Interval { uint64_t offset; uint32_t hits; uint32_t pad; };   // sorted by (bin, time)
struct RunBin { uint32_t ibin, hits; uint64_t offset; };             // FoldCUDA.h:19-24
// one run of the plan recurrence without walking its samples (host_prep.cpp)
uint64_t fold_plan_run(double* phi_io, double pps, double double_nbin, uint64_t nmax, uint32_t* ibin_out);
}  // namespace dspsr_amd

// device plan, double-buffered so that building/uploading the plan of block i+1 never waits for
// the fold kernel of block i (pinned staging => the H2D copies are truly asynchronous)
struct PlanSlot {
  uint32_t* h_bin_start = nullptr;   // pinned
  dspsr_amd::Interval* h_iv = nullptr;          // pinned
  uint32_t* d_bin_start = nullptr;
  dspsr_amd::Interval* d_iv = nullptr;
  size_t bin_cap = 0, iv_cap = 0;
  uint32_t* h_aux = nullptr;         // pinned: time-ordered interval offsets + their index per 1024 samples (segment plan)
  uint32_t* d_aux = nullptr;
  size_t aux_cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  hipEvent_t ready = nullptr;        // the plan's copies (issued on the fold's upload stream) have landed
};

struct dspsr_amd_fold {
  dspsr_amd_ctx* ctx;
  uint32_t nchan = 0, npol = 0, ndim = 0, nbin = 0;
  float* profile = nullptr;     // [chan][pol] rows of nbin*ndim floats, `span` floats apart
  size_t profile_floats = 0;    // floats of the library-owned buffer (0 when the profile is bound to a caller's buffer)
  uint64_t span = 0;            // floats between consecutive (chan, pol) rows
  float* part = nullptr;        // partial profiles of the time segments of a long-run fold (fold.hip, FOLD_LONG_RUN)
  size_t part_floats = 0;
  bool bound = false;           // profile points into the engine-owned device PhaseSeries (dspsr_amd_fold_bind_profile)
  // run-length plan, as CUDA::FoldEngine (FoldCUDA.cu:64-113)
  hipStream_t upload = nullptr;  // plans travel host -> device beside the compute stream (see plan_upload in fold.hip)
  std::vector<dspsr_amd::RunBin> binplan;
  uint32_t current_bin = 0, current_hits = 0, folding_nbin = 0;
  uint64_t ndat_fold = 0;
  PlanSlot slot[2];
  int next_slot = 0;
  std::vector<uint32_t> cursor;
};


// Longest run of the pending plan (samples that go to one phase bin in a row).  Runs of FOLD_LONG_RUN samples or more are
// folded with re-associated sums (fold.hip, k_fold_chunked<., true>); the fused filterbank kernel only has the exact
// time-order fold, so such plans take the separate Detection + Fold launches.
constexpr uint32_t FOLD_LONG_RUN_HOST = 64;
// The fused kernel adds a bin's samples one after the other (exact time order): a run of n samples is a dependent chain of n
// float4 adds on one thread, about 16 cycles each.  Up to this length that still costs less than the detected round trip
// through HBM (headline geometry, ms per block fused / separate: 34-sample runs 4.99 / 5.93, 136: 5.14 / 6.01, 545: 5.42 /
// 5.85, 1090: 6.39 / 5.98; tools/exp_fused_runs.py); beyond it the plan takes the separate launches and the long-run fold.
constexpr uint32_t FOLD_FUSED_MAX_RUN = 640;
static inline uint32_t fold_plan_max_run(const dspsr_amd_fold* f)
{
  uint32_t m = f->current_hits;
  for (const dspsr_amd::RunBin& r : f->binplan) if (r.hits > m) m = r.hits;
  return m;
}

// Fused path (filterbank.hip): turns the pending run-length plan into a per-part plan on the device --
// runs split at multiples of `nkeep`, bucketed by (part, bin), offsets relative to the start of the part.
// start[0..npart] = first active-bin entry of each part, followed (16-byte aligned) by the entries
// {bin, first interval, count << 16 | hits0, offset0} indexing `iv`.  nkeep must be < 65536.  The plan is
// consumed (cleared).
// fold_part_plan_submitted() must be called after the kernels that read the plan have been enqueued.
int fold_build_part_plan(dspsr_amd_fold* f, uint32_t nkeep, uint32_t npart, const uint32_t** d_start,
                         const dspsr_amd::Interval** d_iv, PlanSlot** slot);
// the compute stream waits until the plan in `slot` has landed (call right in front of the first kernel that reads it)
int fold_plan_wait(dspsr_amd_fold* f, PlanSlot* slot);
int fold_part_plan_submitted(dspsr_amd_fold* f, PlanSlot* slot);

// Four-pass fused fold (filterbank.hip k_inv_b<., true>): the pending plan must cover samples [0, ndat) without gaps and
// every interval but the first and the last must hold at least `seg` samples (so that a `seg`-sample run of the last
// inverse pass is cut by at most one phase-bin boundary).  *ok = false: the plan does not qualify and is left pending (the
// caller takes Detection + Fold).  Otherwise the plan is consumed: on the device
//   run_off[0 .. nrun]            start offsets of the time-ordered intervals, run_off[nrun] = ndat
//   blk_first[0 .. ndat/1024]     index of the interval that holds sample 1024*i
//   bin_start / iv                the same intervals bucketed by phase bin (time ordered inside a bin), as for k_fold_chunked
int fold_build_segment_plan(dspsr_amd_fold* f, uint64_t ndat, uint32_t seg, bool* ok, const uint32_t** d_run_off,
                            const uint32_t** d_blk_first, const uint32_t** d_bin_start, const dspsr_amd::Interval** d_iv, PlanSlot** slot);
// profile[chan0 + c][bin] += the segment piece sums of the bin's intervals, in time order (c < nchan).
//   msum: [c][part][tile][t2][2] float4 (ntile = 2^logNt tiles of 2^logTt samples, Mb = 2^logMb runs per tile, run t2 of tile
//   `tile` = output positions tile*Tt + (t2 << logMa) ...); nkeep / nfilt_pos: the kept window of a part
int fold_segment_combine(dspsr_amd_fold* f, const float* msum, uint32_t chan0, uint32_t nchan, uint32_t npart, uint32_t nkeep,
                         uint32_t nfilt_pos, int logTt, int logMa, int logMb, const uint32_t* d_bin_start, const dspsr_amd::Interval* d_iv);
// profile += sum of `nseg` partial profiles (packed [seg][chan][npol][nbin][ndim]) in order: segmented fused launches
int fold_combine_partials(dspsr_amd_fold* f, const float* part, uint32_t nseg, uint32_t chan0, uint32_t nchan);
