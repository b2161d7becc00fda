// Convolving filterbank, inverse pass of the three-pass path (k_inv_chan); instantiated by fb_inv_chan.hip (plain output)
// and fb_inv_chan_fold.hip (fused fold): two translation units so that the two families compile in parallel.
#pragma once
#include "fb_common.h"

namespace dspsr_amd {

// ------------------------------------------------------------------------------------ P3

// T3 output channels (both polarisations) of one part: Hermitian split of spectrum rows s and
// Rr-1-s into the two polarisations (real input), x chirp, inverse M-point FFT, keep window,
// complex output or fused detection.  Columns are (channel, pol) pairs: col = 2*slo + pol, so
// the two butterflies a thread owns are the two polarisations of the same (channel, bin):
// one (a, b, chirp) load serves both and detection needs no cross-lane traffic.
// Items: part fastest, so one XCD re-reads a tile's chirp rows from its L2 for every part.
// FOLD: the detected samples of the tile (T3 channels x nkeep samples, one float4 each) are staged in the
// exchange buffer instead of being written out, and folded at once: thread b owns phase bins b, b + blockDim, ...
// of the tile's channels, loads each touched accumulator from the device profile, adds the samples of the
// bin's intervals one by one in time order and stores it back.  A workgroup processes ALL parts of a tile in
// order and launches are stream ordered, so every (chan, bin) sum has the association order of the CPU loop
// Fold.C:844-852, exactly as the stand-alone fold kernel (fold.hip) -- bit-identical results, without the
// 16 B/sample round trip of the detected time series through HBM.
// EPI: 0 = complex or detected output written by the last stage; 1 (FOLD) = fused fold; 2 = search mode (FbOut kind 5): square-law
// detection + time scrunch of the detected stream (digifil -F N:D, LoadToFil.C:185-222,250-304), staged like the fold's samples and
// reduced by ts_reduce (fb_common.h).  Both walk the parts of a tile in order (the fold's profile, the scrunch's carry).
template <int LOGF, int EPI, int LOGT>
__global__ __launch_bounds__(512) void k_inv_chan(const FbGeom g, const cf* __restrict__ X,
                                                  const cf* __restrict__ kernel, const FbOut out,
                                                  const cf* __restrict__ tw, const uint64_t part0,
                                                  const uint32_t nparts, const uint32_t run)
{
  typedef FftPlan<LOGF> P;
  constexpr bool FOLD = EPI == 1, SEARCH = EPI == 2;
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  uint32_t tid = threadIdx.x;
  const int logT3 = LOGT >= 1 ? LOGT - 1 : g.logT3;
  const int logT = logT3 + 1;
  const uint32_t T = 1u << logT, T3 = 1u << logT3, M = 1u << LOGF, Rr = g.nsub << g.logR;     // (nsub = 3, 5: not a power of two)
  const uint64_t L = (uint64_t)M * Rr;
  const uint32_t nseq = g.real_input ? 1 : g.npol;
  const int logX3 = g.logX3;                    // X layout: element (row, m) at ((row >> logX3)*M + m) << logX3 | row % X3
  const uint32_t X3 = 1u << logX3;
  const uint32_t ntile = g.C >> logT3;
  auto xi = [&](const uint32_t row, const uint32_t m) -> uint64_t {
    return ((((uint64_t)(row >> logX3) << LOGF) + m) << logX3) | (row & (X3 - 1));
  };
  struct Abk { cf a, b; };   // the chirp is fetched at the start of the item (keeps the prefetch at 64 registers)

  // chunk < 0: all elements; otherwise the elements i with i % NCHUNK == chunk (the prefetch of the next tile is
  // issued in NCHUNK groups spread over the transform, see wgfft_stage)
  constexpr int NCHUNK = P::NS + 1;
  // T = 4 columns = 2 channels x 2 polarisations per tile: the two channels' elements are loaded as aligned 16-byte pairs and
  // the halves exchanged between the lane pair.  -3.7 % where the detected or complex output is written; in the fused
  // kernel it cost 1.6 % while the chirp was still loaded per part (round 1) and gains 5.7 % now that it stays in registers
  // (profiles/r02_experiments.txt, item 23)
  constexpr bool PAIR16 = LOGT == 2 && P::G1 == 2;
  const bool pair16 = PAIR16 && g.real_input && logX3 == 1;
  cf special = make_float2(0.f, 0.f);                   // mirror element of bin 0 (pair16 path)
  // a work item = (tile of channels, part of the launch), kept as two 32-bit numbers: a combined 64-bit index costs a
  // software 64-bit division per use (about 300 scalar instructions per tile in the r02c listing)
  struct Item { uint32_t tile, lp; };
  auto fetch = [&](const Item item, Abk (&raw)[PTS / 2], const int chunk) {
    const uint32_t tile = item.tile;
    const cf* __restrict__ X0s = X + (uint64_t)item.lp * nseq * L;
    // element i of a thread's first-stage butterfly is bin m = mb + i*MS of one (channel, pol pair) column, so
    // every address is a base plus a multiple of a wave-uniform step: no per-element index arithmetic, no
    // divergent code between the loads (the m = 0 mirror element, the only irregular one, can only be i = 0)
    constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
    const int64_t step = (int64_t)MS << logX3;
    if constexpr (PAIR16) {
      if (pair16) {
        // Two channels per tile and X3 = 2: the elements of lanes 2j (channel 0) and 2j+1 (channel 1) for the same bin
        // are one aligned 16-byte pair, in both streams.  Lane parity q loads the pairs of the elements i = 2u + q --
        // 16 B per lane, half the load instructions (8-byte-per-lane streams run at 5.6 TB/s, 16-byte ones at 7.1 on this
        // chip, tools/load_width_probe.hip); the halves are exchanged between the two lanes when the tile is consumed.
        const uint32_t q = tid & 1, mb = tid >> 1;
        const uint32_t m0 = mb + q * MS;                                  // bin of element i = q
        const cf* __restrict__ pa2 = X0s + ((((uint64_t)tile << LOGF) + m0) << 1);
        const uint64_t rowb = (uint64_t)((Rr >> 1) - 1 - tile) << LOGF;   // row pair of the mirror rows Rr-1-s
        const cf* __restrict__ pb2 = X0s + ((rowb + (M - m0)) << 1);        // mirror bin M - m of element i = q
#pragma unroll
        for (int u = 0; u < P::R1 / 2; u++) {
          const float4 A = ld_stream((const float4*)(pa2 + (int64_t)u * 4 * MS));
          // bin 0 has its own mirror (loaded below): its pair would lie past the row, read the one before instead
          const float4 B = ld_stream((const float4*)((u == 0 && m0 == 0 ? pb2 - 2 : pb2) - (int64_t)u * 4 * MS));
          raw[2 * u].a = make_float2(A.x, A.y); raw[2 * u + 1].a = make_float2(A.z, A.w);
          raw[2 * u].b = make_float2(B.x, B.y); raw[2 * u + 1].b = make_float2(B.z, B.w);
        }
        const uint32_t s = tile * T3 + q;
        special = ld_stream(mb == 0 ? X0s + xi(s ? Rr - s : 0u, 0) : X0s + xi(Rr - 1 - s, M - mb));
        return;
      }
    }
#pragma unroll
    for (int g2 = 0; g2 < P::G1; g2 += 2) {
      const uint32_t eb = P::G1 * tid + g2;
      const uint32_t slo = (eb & (T - 1)) >> 1, mb = eb >> logT;
      const uint32_t s = tile * T3 + slo;
      const cf* __restrict__ pa = X0s + xi(s, mb);
      const cf* __restrict__ pb = g.real_input ? X0s + xi(Rr - 1 - s, M - mb) : pa + (g.npol == 2 ? L : 0);
      const int64_t stepb = g.real_input ? -step : step;
      const cf* __restrict__ pb0 = (g.real_input && mb == 0) ? X0s + xi(s ? Rr - s : 0u, 0) : pb;
#pragma unroll
      for (int i = 0; i < P::R1; i++) {
        if (chunk >= 0 && i % NCHUNK != chunk) continue;
        Abk r;
        r.a = ld_stream(pa + i * step);
        r.b = ld_stream(i == 0 ? pb0 : pb + i * stepb);
        raw[(g2 / 2) * P::R1 + i] = r;
      }
    }
  };
  // chirp of a tile (fetched at the start of the item: keeps the prefetch at 64 registers)
  auto load_chirp = [&](const Item item, cf (&kk)[PTS / 2]) {
    const uint32_t ktile = item.tile;
    if (kernel) {                                 // uniform; outside the unrolled loads (no per-load branch / vmcnt(0))
      constexpr uint32_t MS = 1u << (LOGF - P::LOGR1);
#pragma unroll
      for (int g2 = 0; g2 < P::G1; g2 += 2) {
        const uint32_t eb = P::G1 * tid + g2;
        const cf* __restrict__ pk = kernel + ((uint64_t)(ktile * T3 + ((eb & (T - 1)) >> 1)) << LOGF) + (eb >> logT);
#pragma unroll
        for (int i = 0; i < P::R1; i++) kk[(g2 / 2) * P::R1 + i] = pk[i * MS];
      }
    } else {
#pragma unroll
      for (int i = 0; i < PTS / 2; i++) kk[i] = make_float2(1.f, 0.f);
    }
  };

  const uint32_t ltw_off = lds_pad(PTS * blockDim.x) + 8;      // behind the exchange buffer (16-byte aligned)
  ltw_fill<LOGF>(lds, ltw_off, tw, tid, blockDim.x);
  // FOLD: two buffers of out.plan_cap plan entries behind the twiddle tables (cf index, 16-byte aligned), followed by
  // a copy of the launch's nparts + 1 part offsets into the plan (PSL_MAX words): a part's entries are then found
  // without a dependent pair of global loads, and are fetched one item ahead (registers) like the tile itself
  const uint32_t plan_off = (ltw_off + ltw_entries_dev<LOGF>() + 1) & ~1u;
  uint32_t* psl = nullptr;
  const uint4* __restrict__ fent_all = nullptr;
  bool use_psl = false;
  if constexpr (FOLD) {
    psl = (uint32_t*)&lds[plan_off + 4 * out.plan_cap];
    fent_all = (const uint4*)(out.pstart + ((out.nparts_plan + 1 + 3) & ~3u));
  }
  uint32_t jt = 0;                                              // tiles done by this workgroup
  Item item, next;
  uint32_t j = 0;
  // FOLD: workgroup b takes tiles b, b + grid, ... and walks the parts of each in order.
  // Segmented (out.nseg > 1, geometries with fewer channel tiles than compute units): the parts of the launch are cut
  // into nseg runs; workgroup (lane, seg) walks the parts of run `seg` for tiles lane, lane + ntg, ...  Run 0 adds onto
  // the profile (it continues the sums of earlier launches in time order), the others onto zeroed partial profiles that
  // are added to the profile, in run order, after the launch -- the sums of a launch are re-associated per run.
  uint32_t fold_b = blockIdx.x;
  const uint32_t fnseg = FOLD && out.nseg > 1 ? out.nseg : 1u;
  const uint32_t fntg = gridDim.x / fnseg;                         // workgroups per run (host: gridDim.x % nseg == 0)
  const uint32_t fseg = fnseg > 1 ? fold_b / fntg : 0u;
  const uint32_t fpps = (nparts + fnseg - 1) / fnseg;              // parts per run
  const uint32_t fp0 = fseg * fpps;
  const uint32_t fnp = fp0 >= nparts ? 0u : (nparts - fp0 < fpps ? nparts - fp0 : fpps);
  if (fnseg > 1) fold_b -= fseg * fntg;
  if constexpr (FOLD) {
    if (fnp == 0) return;
    // the offsets of the parts THIS workgroup walks (its run of a segmented launch: launches of up to 256 parts are cut into
    // runs of at most FB_PSL_MAX - 1; the whole launch's offsets did not fit, and every entry and accumulator of such launches
    // -- the sub-band and -F 256:D geometries -- then came from global memory in the fold phase), psl[lp - fp0]
    use_psl = out.plan_cap > 0 && fnp + 1 <= FB_PSL_MAX;
    if (use_psl) {
      for (uint32_t q = tid; q <= fnp; q += blockDim.x) psl[q] = out.pstart[part0 + fp0 + q];
      __syncthreads();
    }
    // tiles that share an X layout block (2^(logX3-logT3) of them) go to blocks b, b+8, ... : one XCD under the
    // observed round-robin placement, at the same time, so the block's lines are fetched once (speed only)
    const int lr = logX3 - logT3;
    if (fnseg == 1 && lr > 0 && (gridDim.x & ((8u << lr) - 1)) == 0)
      fold_b = ((((fold_b >> (3 + lr)) << 3) | (fold_b & 7)) << lr) | ((fold_b >> 3) & ((1u << lr) - 1));
  }
  // Without the fold the order of the items is free.  When every workgroup gets the same number of tiles it also walks the
  // parts of a tile one after the other, so that the tile's chirp stays in registers (one chirp read per launch, not per
  // part); otherwise the items are dealt XCD-wise as in the other passes.
  const bool tile_major = EPI != 0 || (ntile >= gridDim.x && ntile % gridDim.x == 0);
  auto next_item = [&](const uint32_t jj, Item& it) -> bool {
    if (EPI != 0 || tile_major) {
      const uint32_t q = jj / fnp;                     // (32-bit; jj counts this workgroup's items)
      it.tile = fold_b + q * fntg;
      it.lp = fp0 + (jj - q * fnp);
      return it.tile < ntile;
    } else {
      // XCD dealing as persistent_item (wgfft.h) with runs of `run` items; run == nparts (the default) makes the run
      // index the tile and the position in the run the part, without a division by nparts
      const uint32_t grid = gridDim.x, b = blockIdx.x;
      uint32_t hi, lo;
      if (grid & 7) {
        const uint32_t lin = b + jj * grid;
        hi = lin / run; lo = lin - hi * run;
      } else {
        const uint32_t q = jj * (grid >> 3) + (b >> 3);
        const uint32_t qr = q / run;
        hi = qr * 8 + (b & 7); lo = q - qr * run;
      }
      if (run == nparts) { it.tile = hi; it.lp = lo; }
      else { const uint32_t lin = hi * run + lo; it.tile = lin / nparts; it.lp = lin - it.tile * nparts; }
      return it.tile < ntile;                          // (lp < nparts by construction)
    }
  };
  if (!next_item(j, item)) return;
  Abk raw[PTS / 2];
  fetch(item, raw, -1);
  cf kk[PTS / 2];                       // chirp of the current tile
  uint32_t kk_tile = ~0u;
  // FOLD: plan entry of this thread for the item about to be processed (tid < number of active bins of the part)
  uint32_t fe0_cur = 0, fn_cur = 0;
  auto plan_fetch = [&](const Item it) {
    if constexpr (FOLD) {
      const uint32_t lp = it.lp;
      if (use_psl) { fe0_cur = psl[lp - fp0]; fn_cur = psl[lp - fp0 + 1] - fe0_cur; }
      else { fe0_cur = out.pstart[part0 + lp]; fn_cur = out.pstart[part0 + lp + 1] - fe0_cur; }
    }
  };
  // The active-bin entries of a part (16 bytes each, at most plan_cap <= blockDim of them) go from global memory STRAIGHT into
  // their LDS buffer (global_load_lds_dwordx4: lane l of a wave lands at the wave's base + 16*l), asynchronously and
  // without passing through registers.  Round 2 fetched them into a register at the top of the tile and stored them to
  // LDS: the kernel sits at 256 registers, the value was spilled to scratch, and the ISA read `s_waitcnt vmcnt(0);
  // global_load; s_waitcnt vmcnt(0); scratch_store; ...; scratch_load; s_waitcnt vmcnt(0); ds_write` -- two exposed memory
  // round trips on the two waves everyone then waits for at the first barrier.  The entries of the NEXT item are now
  // requested in the middle of the current tile's transform (behind a barrier that the previous readers of that buffer
  // have passed) and are waited for, together with the prefetched tile, at the top of the next one.
  // (needs a second stage: the request for the next item is issued from wgfft's `mid` hook behind the first exchange barrier;
  //  single-stage transforms read their entries from global memory)
  const bool plan_dma_ok = FOLD && FftPlan<LOGF>::NS >= 2 && use_psl;
  auto plan_dma = [&](const Item it, const uint32_t buf) {
    if constexpr (FOLD) {
      const uint32_t fe0 = psl[it.lp - fp0], fn = psl[it.lp - fp0 + 1] - fe0;
      if (fn <= out.plan_cap && tid < fn)
        lds_dma_b128((const void*)(fent_all + fe0 + tid), lds_byte_addr((const uint4*)&lds[plan_off] + buf * out.plan_cap + (tid & ~63u)));
    }
  };
  if (plan_dma_ok) plan_dma(item, 0);

  FB_ST_BEGIN(3);
  for (;;) {
    asm volatile("" : "+v"(tid));   // per-tile index math stays inside the loop (see wgfft)
    cx2 x[NPAIR];
    FB_ST(3, 0);                     // (waits for the prefetched tile first)
    {
      // FOLD: a workgroup walks the parts of ITS tile, so consecutive items share the chirp rows: they are loaded when
      // the tile changes and stay in registers (the load and its latency were 22 % of the tile, profiles/r02c_*)
      if (item.tile != kk_tile) {
        load_chirp(item, kk);
        kk_tile = item.tile;
      }
      if constexpr (FOLD) {
        // this part's active-bin entries travel with the chirp loads and are parked in LDS (double buffered: slower
        // waves may still be folding the previous tile from the other half); their offsets come from the LDS copy
        plan_fetch(item);
        // the entries requested during the previous tile (or in front of the loop) have landed once every older load has
        // -- they were issued half a tile ago, behind the prefetch of this tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (PAIR16) {
        if (pair16) {                       // hand the other channel's halves of the 16-byte pairs to the neighbour lane
          const bool q = tid & 1;
          auto swap1 = [](const cf v) {     // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
            return make_float2(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, false)),
                               __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, false)));
          };
#pragma unroll
          for (int u = 0; u < P::R1 / 2; u++) {
            const cf alo = raw[2 * u].a, ahi = raw[2 * u + 1].a;            // (channel 0, channel 1) of element 2u + q
            const cf aown = q ? ahi : alo, arecv = swap1(q ? alo : ahi);
            raw[2 * u].a = q ? arecv : aown;
            raw[2 * u + 1].a = q ? aown : arecv;
            const cf blo = raw[2 * u].b, bhi = raw[2 * u + 1].b;            // mirror rows: (channel 1, channel 0)
            const cf bown = q ? blo : bhi, brecv = swap1(q ? bhi : blo);
            raw[2 * u].b = q ? brecv : bown;
            raw[2 * u + 1].b = q ? bown : brecv;
          }
          if ((tid >> 1) == 0) raw[0].b = special;
        }
      }
      // (the uniform real/complex choice is made once, outside the unrolled loops: no branch per element)
      if (g.real_input) {
#pragma unroll
        for (int q = 0; q < PTS / 2; q++) {
          const Abk r = raw[q];
          // W[k] = X0[k] + i X1[k] ; conj(W[L-k]) = X0[k] - i X1[k]
          const cf x0 = make_float2(0.5f * (r.a.x + r.b.x), 0.5f * (r.a.y - r.b.y));
          const cf x1 = make_float2(0.5f * (r.a.y + r.b.y), 0.5f * (r.b.x - r.a.x));
          x[q] = cmuls(make_cx2(x0, x1), kk[q]);          // Response::operate, Response.C:429-441
        }
      } else {
        const bool two = g.npol == 2;
#pragma unroll
        for (int q = 0; q < PTS / 2; q++) {
          const Abk r = raw[q];
          x[q] = cmuls(make_cx2(r.a, two ? r.b : make_float2(0.f, 0.f)), kk[q]);
        }
      }
    }
    FB_ST(3, 1);
    const bool more = next_item(++j, next);
    // One burst, and unconditional: the last item of a workgroup is fetched again and dropped.  Under `if (more)` the generic
    // (8-byte) form's loads went to fresh registers and the copies into the loop-carried `raw` sat at the end of the conditional
    // block behind `s_waitcnt vmcnt(16) ... (0)` -- the prefetch was waited for at once (profiles/r04_experiments.txt item 13;
    // what rounds 1-3 read as "the wave time goes to ISSUING the 8-byte loads").  The 16-byte pair form of the headline
    // geometry was not affected.
    fetch(more ? next : item, raw, -1);
    FB_ST(3, 2);

    const uint32_t tile = item.tile;
    const uint64_t part = part0 + item.lp;
    [[maybe_unused]] TsPart tsp = {0, 0, 0, 0};
    [[maybe_unused]] float ts_carry_pre = 0.f;
    if constexpr (SEARCH) tsp = ts_part(out, (uint32_t)part, g.nkeep);
    // FOLD: the tile's detected samples are staged UNPADDED, channel after channel (16 bytes per sample), so that the
    // samples of a phase bin's run are read at constant offsets from one base (the padded image cost four integer
    // instructions per sample in a phase that only three of eight waves work in).  The channel stride is nkeep rounded up
    // so that the T3 channels a quarter wave writes at once fall on different LDS banks.
    // (many channels of a short transform: the rounded stride would not fit the exchange buffer -- 2*T3*nkeep words always do)
    const uint32_t fcr = (16u >> logT3) & 15u, fcs_r = ((g.nkeep + 15u - fcr) & ~15u) + fcr;
    const uint32_t fcs = ((2u * fcs_r) << logT3) <= PTS * blockDim.x ? fcs_r : g.nkeep;
    auto store = [&](const uint32_t col, const uint32_t p, const uint32_t pstride, auto& v) {
      constexpr int R = sizeof(v) / sizeof(v[0]);
      if constexpr (FOLD) {
        const uint32_t slo = col >> 1;
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;           // outside the kept window (negative t wraps)
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          *(float4*)&lds[2 * (slo * fcs + (uint32_t)t)] = make_float4(r[0], r[1], r[2], r[3]);
        }
        return;
      }
      if constexpr (SEARCH) {
        const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t t = t0 + (int32_t)(k * pstride);
          if ((uint32_t)t >= g.nkeep) continue;
          // (one polarisation: the pair's second half is not a signal -- for real input the other output of the Hermitian split,
          //  zero but for rounding, 1e-7 of the spectrum's scale: its power moved the sum by an ulp where a sample is small,
          //  tests/fuzz_search.py 300 702 case 68)
          ts_stage((float*)lds, out, tsp, col >> 1, (uint32_t)t, cx2_lo(v[k]), g.npol == 1 ? make_float2(0.f, 0.f) : cx2_hi(v[k]));
        }
        return;
      }
      if (out.kind == 0) return;
      const uint32_t chan = out.chan0 + tile * T3 + (col >> 1);
      float* __restrict__ row = out.base + chan * out.chan_stride;
      // output sample of element k: t0 + k*pstride (kept when 0 <= t < nkeep); the addresses are a base plus a
      // multiple of a wave-uniform step
      const int32_t t0 = (int32_t)p - (int32_t)g.nfilt_pos;
      float2* __restrict__ o2 = (float2*)(row + part * out.part_step) + t0;
      float4* __restrict__ o4 = (float4*)row + ((int64_t)(part * g.nkeep) + t0);
      // the output kind / layout is uniform: chosen once, outside the unrolled element loop (no branch per element)
      if (out.kind == 1) {
        const bool two = g.npol == 2;
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float2* o = o2 + k * pstride;
          st_stream(o, cx2_lo(v[k]));
          if (two) st_stream((float2*)((float*)o + out.pol_stride), cx2_hi(v[k]));
        }
      } else if (out.ndim == 4) {
#pragma unroll
        for (int k = 0; k < R; k++) {
          if ((uint32_t)(t0 + (int32_t)(k * pstride)) >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          st_stream(o4 + k * pstride, make_float4(r[0], r[1], r[2], r[3]));
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
          const int32_t ts = t0 + (int32_t)(k * pstride);
          if ((uint32_t)ts >= g.nkeep) continue;
          float r[4];
          detect4(cx2_lo(v[k]), cx2_hi(v[k]), out.state, r);
          const uint64_t idat = part * g.nkeep + (uint32_t)ts;
          if (out.ndim == 2) {
            st_stream(&((float2*)row)[idat], make_float2(r[0], r[1]));
            st_stream(&((float2*)(row + out.pol_stride))[idat], make_float2(r[2], r[3]));
          } else {
            row[idat] = r[0];
            row[out.pol_stride + idat] = r[1];
            row[2 * out.pol_stride + idat] = r[2];
            row[3 * out.pol_stride + idat] = r[3];
          }
        }
      }
    };
    // FOLD: active phase bins of this part: entries {bin, first interval, count<<16 | hits0, offset0}
    // (fold_internal.h), copied to LDS at the start of the tile when they fit; one (entry, channel) accumulator per
    // work item.  The accumulator of a thread's first work item (nearly always its only one) is requested in the
    // middle of the transform -- behind two barriers, so the previous part's stores of this workgroup are visible --
    // and arrives while the last stage runs, instead of costing a memory round trip in the fold phase.
    uint32_t f_e0 = 0, f_nact = 0;
    const uint4* __restrict__ ent = nullptr;
    const uint4* planl = nullptr;
    bool in_lds = false;
    uint4 en_pre = make_uint4(0, 0, 0, 0);
    float4 acc_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr bool PRE = FOLD && FftPlan<LOGF>::NS >= 2;
    if constexpr (FOLD) {
      f_e0 = fe0_cur;
      f_nact = fn_cur;
      ent = fent_all + f_e0;
      planl = (const uint4*)&lds[plan_off] + (jt & 1) * out.plan_cap;
      in_lds = plan_dma_ok && f_nact <= out.plan_cap;
    }
    // accumulator of (work item w, phase bin b): one float4, or -- profile of npol 2 x ndim 2 -- a float2 in each of the
    // channel's two rows; the partial profiles of a segmented launch are packed in the same shape (rows of nbin bins)
    const bool planes2 = FOLD && out.prof_planes == 2;                       // uniform
    const uint64_t plane = fseg == 0 ? out.plane_stride : 2ull * out.nbin;   // floats between the two rows of a channel
    auto acc_row = [&](const uint32_t w) -> float* {
      const uint32_t cl = tile * T3 + (w & (T3 - 1));              // channel within this input channel's sub-band
      return (float*)(fseg == 0 ? (float4*)out.base + (uint64_t)(out.chan0 + cl) * out.prof_span4
                                : (float4*)out.part + ((uint64_t)(fseg - 1) * g.C + cl) * out.nbin);
    };
    auto acc_load = [&](float* row, const uint32_t b) -> float4 {
      if (planes2) {
        const float2 u = *(const float2*)(row + 2 * b), v = *(const float2*)(row + plane + 2 * b);
        return make_float4(u.x, u.y, v.x, v.y);
      }
      return *(const float4*)(row + 4 * b);
    };
    auto acc_store = [&](float* row, const uint32_t b, const float4 a) {
      if (planes2) {
        *(float2*)(row + 2 * b) = make_float2(a.x, a.y);
        *(float2*)(row + plane + 2 * b) = make_float2(a.z, a.w);
      } else {
        *(float4*)(row + 4 * b) = a;
      }
    };
    auto mid = [&](const int phase) {
      if constexpr (SEARCH) {
        // the open output sample's partial sum (written by this workgroup at the end of the previous part, two barriers ago)
        const uint32_t npo = out.state == DSPSR_AMD_PPQQ ? 2u : 1u;
        // (behind the tile's first barrier: that store has been waited for; single-stage transforms load it in ts_reduce)
        if (FftPlan<LOGF>::NS >= 2 && phase == 2 && tsp.phi && tid < (npo << logT3))
          ts_carry_pre = ts_carry_load(out, (out.chan0 + tile * T3 + tid / npo) * npo + tid % npo);
      }
      if constexpr (PRE) {
        // only when the part's plan entries are in LDS: the accumulator's address then depends on an LDS read alone.  With
        // the entry possibly coming from global memory (a select, or two branches that the compiler merges again) the
        // load below sat behind s_waitcnt vmcnt(0) -- a wait for the whole prefetch of the next tile, in the middle of the
        // transform, on exactly the three waves that also fold
        if (phase == 2 && in_lds && tid < (f_nact << logT3)) {
          en_pre = planl[tid >> logT3];
          acc_pre = acc_load(acc_row(tid), en_pre.x);
        }
        // every wave is past this tile's first exchange barrier, i.e. has left the previous tile's fold: the other plan
        // buffer is free for the entries of the next item
        if (phase == 2 && plan_dma_ok && more) plan_dma(next, (jt + 1) & 1);
      }
    };
    wgfft<LOGF, +1, EPI != 0>(lds, ltw_off, tid, logT, x, store, mid);
    FB_ST(3, 3);
    if constexpr (SEARCH) {
      __syncthreads();                       // the tile's detected samples are staged
      const uint32_t npo = out.state == DSPSR_AMD_PPQQ ? 2u : 1u;
      ts_reduce((const float*)lds, out, tsp, npo << logT3, ts_carry_pre, FftPlan<LOGF>::NS >= 2, tid, blockDim.x, [&](const uint32_t slo) { return tile * T3 + slo; });
      // (the barrier in front of the next tile's first exchange write also ends this read phase)
    }
    if constexpr (FOLD) {
      __syncthreads();                       // the tile's detected samples are staged
      // the samples of an interval are fetched from LDS eight at a time (independent loads) and then added one after
      // the other, so the sum keeps the time order
      const bool pre = PRE && in_lds;
      for (uint32_t w = tid; w < (f_nact << logT3); w += blockDim.x) {
        const uint32_t slo = w & (T3 - 1);
        uint4 en;
        float4 acc;
        if (pre && w == tid) {
          en = en_pre;
          acc = acc_pre;
        } else {
          en = in_lds ? planl[w >> logT3] : ent[w >> logT3];
          acc = acc_load(acc_row(w), en.x);
        }
        const uint32_t nint = en.z >> 16;
        float* __restrict__ pp = acc_row(w);
        uint32_t off = en.w, hits = en.z & 0xffffu;
        for (uint32_t i = 0;;) {
          const float4* __restrict__ src = (const float4*)&lds[2 * (slo * fcs + off)];     // consecutive samples: constant offsets
          uint32_t h = 0;
          for (; h + 8 <= hits; h += 8) {
            float4 sm[8];
#pragma unroll
            for (int q = 0; q < 8; q++) sm[q] = src[h + q];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc.x += sm[q].x; acc.y += sm[q].y; acc.z += sm[q].z; acc.w += sm[q].w; }
          }
          for (; h < hits; h++) {
            const float4 sm = src[h];
            acc.x += sm.x; acc.y += sm.y; acc.z += sm.z; acc.w += sm.w;
          }
          if (++i >= nint) break;
          const Interval iv = out.piv[en.y + i];           // further intervals of the bin in this part (rare)
          off = (uint32_t)iv.offset; hits = iv.hits;
        }
        acc_store(pp, en.x, acc);
      }
      // the barrier in front of the next tile's first exchange write also ends this read phase
    }
    FB_ST(3, 4);
    FB_ST_TILE(3, 5);
    if (!more) break;
    item = next;
    jt++;
  }
  FB_ST_END(3);
}


}  // namespace dspsr_amd
