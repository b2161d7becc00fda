// The ONE exchange of the path: the sub-integration dump over RCCL (xGMI), behind the C-ABI so that DSPSR's C++ host can
// call it (it cannot call torch.distributed).
//
// Reference: the dump point is dsp::Subint<Fold>::transformation (Signal/Pulsar/dsp/Subint.h:291-303); what merging two
// PhaseSeries means is PhaseSeries::combine (Signal/Pulsar/PhaseSeries.C:442-484: profile += other, hits += other.hits,
// integration_length and ndat_total add); today the reference merges the pieces of its threads on the host
// (Signal/General/MultiThread.C:274-379, UnloaderShare).
//
//   DSPSR_AMD_REDUCE_SUM    time-slice replicas (single-channel input): ONE ncclReduce(SUM) of ONE packed buffer of doubles
//                           [profile | hits | ndat_total | integration_length].  Integers below 2^53 add exactly in double;
//                           the profile sums are rounded to float once on the root (the reference adds floats pairwise in
//                           thread-arrival order: equal to rounding, ours does not depend on the order).
//   DSPSR_AMD_REDUCE_GATHER sub-band shards: every rank's slice is delivered to the root in rank order -- an
//                           ncclGather of the slices (8 MiB each at cfg4), not a zero-padded full-band SUM; hits[],
//                           integration_length and ndat_total are identical on all ranks (one channel-independent bin
//                           plan, Fold.C:744-787) and are the root's own; `check_hits` adds a MIN/MAX all-reduce of hits[]
//                           to the same group and reports whether all ranks agree.
//
// start() snapshots the profile on the context's (compute) stream -- a device-to-device pack into the communicator's
// staging buffer -- so the caller may zero the profile and launch the next block at once; the collective and the copy to
// pinned host memory run on the communicator's own stream behind an event.  finish() waits for them.
//
// RCCL is opened with dlopen at the first communicator (no link-time dependency: the library loads, and every other
// entry point works, where no RCCL is installed; a process that already holds an RCCL -- PyTorch's -- shares it).
#include <dlfcn.h>
#include <string.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Built where the RCCL headers are not installed: the few types and enumerators the dlopen'ed entry points take, as rccl.h
// declares them (NCCL's stable ABI).  The library then still builds, and comm_* report at run time whether a librccl loads.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}
#endif

#include "engine_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = "";
};

char g_rccl_path[1024] = "";     // dspsr_amd_comm_set_library

Rccl* rccl()
{
  static Rccl r;
  static std::mutex mtx;
  std::lock_guard<std::mutex> lock(mtx);
  if (r.handle) return &r;
  const char* names[] = {g_rccl_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n[0]) continue;
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) { snprintf(r.why, sizeof r.why, "dlopen(librccl.so.1): %s", dlerror()); return &r; }
  bool ok = true;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.handle, name);
    if (!p) { ok = false; snprintf(r.why, sizeof r.why, "librccl: symbol %s missing", name); }
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.Gather = (decltype(r.Gather))sym("ncclGather");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(r.handle); r.handle = nullptr; }
  return &r;
}

// profile rows (span floats apart) -> packed staging buffer, as double (SUM) or float (GATHER)
template <typename T>
__global__ void k_pack_rows(const float* __restrict__ prof, const uint64_t span, const uint64_t row_floats, const uint64_t n,
                            T* __restrict__ out)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / row_floats, k = i - r * row_floats;
    out[i] = (T)prof[r * span + k];
  }
}

// SUM mode, root: the reduced doubles back to float (rounded once), so that half the bytes cross to the host and no host loop
// touches the profile
__global__ void k_unpack_sum(const double* __restrict__ in, float* __restrict__ out, const uint64_t n)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

}  // namespace

struct dspsr_amd_comm {
  dspsr_amd_ctx* ctx = nullptr;
  Rccl* lib = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  hipStream_t stream = nullptr;          // the collective's own stream
  hipEvent_t ready = nullptr, done = nullptr;
  void* send = nullptr; size_t send_cap = 0;      // device staging
  void* recv = nullptr; size_t recv_cap = 0;      // device, root only
  void* host = nullptr; size_t host_cap = 0;      // pinned: meta going out, result coming in
  uint32_t* hmm = nullptr;                        // device: MIN then MAX of hits over the ranks (check_hits)
  size_t hmm_cap = 0;
  // the exchange in flight
  bool pending = false;
  int mode = 0, root = 0, check = 0;
  uint64_t n = 0;                                 // profile floats of this rank
  uint32_t nbin = 0;
  std::vector<uint32_t> hits;                     // GATHER: the rank's own values (the root's are the result)
  double length = 0.0;
  uint64_t ndat_total = 0;
  size_t tail_off = 0;                            // check_hits: where the MIN / MAX of hits[] arrive in `host`
  size_t meta_off = 0;                            // SUM, root: where the reduced counters (doubles) arrive in `host`
};

static int comm_fail(dspsr_amd_comm* c, int code, const char* what, ncclResult_t r)
{
  return ctx_fail(c->ctx, code, "%s: %s", what, c->lib && c->lib->GetErrorString ? c->lib->GetErrorString(r) : "RCCL error");
}

static bool reserve_dev(void** p, size_t* cap, size_t need)
{
  if (need <= *cap) return true;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  if (hipMalloc(p, need) != hipSuccess) return false;
  *cap = need;
  return true;
}

extern "C" int dspsr_amd_comm_set_library(const char* path)
{
  if (!path || strlen(path) >= sizeof g_rccl_path) return DSPSR_AMD_EINVAL;
  strcpy(g_rccl_path, path);
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_comm_unique_id(void* id_out)
{
  if (!id_out) return DSPSR_AMD_EINVAL;
  Rccl* lib = rccl();
  if (!lib->handle) return DSPSR_AMD_ESTATE;
  ncclUniqueId id;
  if (lib->GetUniqueId(&id) != ncclSuccess) return DSPSR_AMD_EHIP;
  static_assert(sizeof(id) == DSPSR_AMD_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof id);
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_comm_create(dspsr_amd_ctx* ctx, int nranks, int rank, const void* unique_id, dspsr_amd_comm** out)
{
  if (!ctx || !out || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return DSPSR_AMD_EINVAL;
  *out = nullptr;
  Rccl* lib = rccl();
  if (!lib->handle) return ctx_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_comm_create: RCCL not available: %s", lib->why);
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_comm_create: hipSetDevice failed");
  dspsr_amd_comm* c = new dspsr_amd_comm;
  c->ctx = ctx; c->lib = lib; c->nranks = nranks; c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  const ncclResult_t r = lib->CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) { const int rc = comm_fail(c, DSPSR_AMD_EHIP, "dspsr_amd_comm_create: ncclCommInitRank", r); delete c; return rc; }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    dspsr_amd_comm_destroy(c);
    return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_comm_create: stream / event creation failed");
  }
  *out = c;
  return DSPSR_AMD_OK;
}

extern "C" void dspsr_amd_comm_destroy(dspsr_amd_comm* c)
{
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)c->lib->CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->hmm) (void)hipFree(c->hmm);
  if (c->host) (void)hipHostFree(c->host);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int dspsr_amd_comm_rank(const dspsr_amd_comm* c) { return c ? c->rank : -1; }
extern "C" int dspsr_amd_comm_size(const dspsr_amd_comm* c) { return c ? c->nranks : 0; }

extern "C" int dspsr_amd_reduce_profiles_start(dspsr_amd_comm* c, int mode, int root, const float* profile_dev, uint64_t span_floats,
                                               uint64_t nrow, uint64_t row_floats, const uint32_t* hits_host, uint32_t nbin,
                                               double integration_length, uint64_t ndat_total, int check_hits)
{
  if (!c || !profile_dev || !hits_host || !nrow || !row_floats || span_floats < row_floats || root < 0 || root >= c->nranks)
    return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = c->ctx;
  if (mode != DSPSR_AMD_REDUCE_SUM && mode != DSPSR_AMD_REDUCE_GATHER)
    return ctx_fail(ctx, DSPSR_AMD_EINVAL, "dspsr_amd_reduce_profiles_start: unknown mode %d", mode);
  if (c->pending) return ctx_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_reduce_profiles_start: the previous exchange was not finished");
  // the current device is a per-thread setting: a caller thread that has not selected this context's GPU (a fresh helper thread
  // starts on device 0) would place the staging buffers below on another device than the stream and the communicator
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_start: hipSetDevice failed");
  const uint64_t n = nrow * row_floats;
  const bool sum = mode == DSPSR_AMD_REDUCE_SUM, is_root = c->rank == root;
  const size_t nmeta = (size_t)nbin + 2;                                   // SUM: hits, ndat_total, integration_length as doubles
  const size_t send_bytes = sum ? (n + nmeta) * sizeof(double) : n * sizeof(float);
  const size_t recv_bytes = !is_root ? 0 : (sum ? send_bytes : (size_t)c->nranks * n * sizeof(float));
  // pinned host buffer: [merged profile, floats (root)] [merged counters, doubles (root, SUM)] [this rank's counters going out
  // (SUM)] [MIN / MAX of hits[] (check_hits)]
  const size_t prof_bytes = !is_root ? 0 : (((sum ? n : (size_t)c->nranks * n) * sizeof(float) + 7) & ~(size_t)7);
  const size_t meta_off = prof_bytes, stage_off = meta_off + (is_root && sum ? nmeta * sizeof(double) : 0);
  const size_t tail_off = stage_off + (sum ? nmeta * sizeof(double) : 0);
  const size_t host_bytes = tail_off + 2 * (size_t)nbin * sizeof(uint32_t);
  if (!reserve_dev(&c->send, &c->send_cap, send_bytes) || (recv_bytes && !reserve_dev(&c->recv, &c->recv_cap, recv_bytes)) ||
      (check_hits && !reserve_dev((void**)&c->hmm, &c->hmm_cap, 2 * (size_t)nbin * sizeof(uint32_t))))
    return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_reduce_profiles_start: staging allocation failed");
  if (host_bytes > c->host_cap) {
    (void)hipStreamSynchronize(c->stream);
    if (c->host) (void)hipHostFree(c->host);
    c->host = nullptr; c->host_cap = 0;
    if (hipHostMalloc(&c->host, host_bytes) != hipSuccess)
      return ctx_fail(ctx, DSPSR_AMD_ENOMEM, "dspsr_amd_reduce_profiles_start: pinned allocation failed");
    c->host_cap = host_bytes;
  }
  // ---- compute stream: snapshot of the profile (and, SUM, of the counters) into the staging buffer
  uint32_t gx = (uint32_t)((n + 255) / 256);
  if (gx > 8 * ctx->ncu) gx = 8 * ctx->ncu;
  hipError_t e = hipSuccess;
  if (sum) {
    hipLaunchKernelGGL(k_pack_rows<double>, dim3(gx), dim3(256), 0, ctx->stream, profile_dev, span_floats, row_floats, n, (double*)c->send);
    double* m = (double*)((char*)c->host + stage_off);
    for (uint32_t b = 0; b < nbin; b++) m[b] = (double)hits_host[b];
    m[nbin] = (double)ndat_total;
    m[nbin + 1] = integration_length;
    e = hipMemcpyAsync((double*)c->send + n, m, nmeta * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  } else {
    hipLaunchKernelGGL(k_pack_rows<float>, dim3(gx), dim3(256), 0, ctx->stream, profile_dev, span_floats, row_floats, n, (float*)c->send);
  }
  if (e == hipSuccess && check_hits) {
    // hits[] of this rank, twice (the all-reduces work in place: MIN in the first copy, MAX in the second); staged in the
    // tail of the pinned buffer, which the result does not reach
    uint32_t* hh = (uint32_t*)((char*)c->host + tail_off);
    memcpy(hh, hits_host, nbin * sizeof(uint32_t));
    memcpy(hh + nbin, hits_host, nbin * sizeof(uint32_t));
    e = hipMemcpyAsync(c->hmm, hh, 2 * (size_t)nbin * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(c->ready, ctx->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ready, 0);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_start: %s", hipGetErrorString(e));
  // ---- the communicator's stream: ONE group
  Rccl* L = c->lib;
  ncclResult_t r = L->GroupStart();
  if (r == ncclSuccess)
    r = sum ? L->Reduce(c->send, c->recv, n + nmeta, ncclDouble, ncclSum, root, c->comm, c->stream)
            : L->Gather(c->send, c->recv, n, ncclFloat, root, c->comm, c->stream);
  if (r == ncclSuccess && check_hits) r = L->AllReduce(c->hmm, c->hmm, nbin, ncclUint32, ncclMin, c->comm, c->stream);
  if (r == ncclSuccess && check_hits) r = L->AllReduce(c->hmm + nbin, c->hmm + nbin, nbin, ncclUint32, ncclMax, c->comm, c->stream);
  const ncclResult_t r2 = L->GroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) return comm_fail(c, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_start: RCCL", r);
  if (is_root && sum) {
    // the staging buffer is free once the reduce has read it: the floats of the merged profile go there, then to the host
    hipLaunchKernelGGL(k_unpack_sum, dim3(gx), dim3(256), 0, c->stream, (const double*)c->recv, (float*)c->send, n);
    e = hipMemcpyAsync(c->host, c->send, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync((char*)c->host + meta_off, (const double*)c->recv + n, nmeta * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  } else if (is_root) {
    e = hipMemcpyAsync(c->host, c->recv, recv_bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess && check_hits)
    e = hipMemcpyAsync((char*)c->host + tail_off, c->hmm, 2 * (size_t)nbin * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipEventRecord(c->done, c->stream);
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_start: %s", hipGetErrorString(e));
  c->pending = true;
  c->mode = mode; c->root = root; c->check = check_hits; c->n = n; c->nbin = nbin;
  c->tail_off = tail_off;
  c->meta_off = meta_off;
  c->hits.assign(hits_host, hits_host + nbin);
  c->length = integration_length;
  c->ndat_total = ndat_total;
  return DSPSR_AMD_OK;
}

extern "C" int dspsr_amd_reduce_profiles_finish(dspsr_amd_comm* c, float* profile_host, uint32_t* hits_host, double* integration_length,
                                                uint64_t* ndat_total, int* hits_identical)
{
  if (!c) return DSPSR_AMD_EINVAL;
  dspsr_amd_ctx* ctx = c->ctx;
  if (!c->pending) return ctx_fail(ctx, DSPSR_AMD_ESTATE, "dspsr_amd_reduce_profiles_finish: no exchange in flight");
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_finish: hipSetDevice failed");
  const hipError_t e = hipEventSynchronize(c->done);
  c->pending = false;
  if (e != hipSuccess) return ctx_fail(ctx, DSPSR_AMD_EHIP, "dspsr_amd_reduce_profiles_finish: %s", hipGetErrorString(e));
  const uint64_t n = c->n;
  const uint32_t nbin = c->nbin;
  if (hits_identical) {
    *hits_identical = 1;
    if (c->check) {
      const uint32_t* hh = (const uint32_t*)((const char*)c->host + c->tail_off);        // MIN over the ranks, then MAX
      for (uint32_t b = 0; b < nbin; b++) if (hh[b] != hh[nbin + b]) { *hits_identical = 0; break; }
    }
  }
  if (c->rank != c->root) return DSPSR_AMD_OK;
  if (c->mode == DSPSR_AMD_REDUCE_SUM) {
    const double* d = (const double*)((const char*)c->host + c->meta_off);
    if (profile_host) memcpy(profile_host, c->host, n * sizeof(float));
    if (hits_host) for (uint32_t b = 0; b < nbin; b++) hits_host[b] = (uint32_t)d[b];
    if (ndat_total) *ndat_total = (uint64_t)d[nbin];
    if (integration_length) *integration_length = d[nbin + 1];
  } else {
    if (profile_host) memcpy(profile_host, c->host, (size_t)c->nranks * n * sizeof(float));
    if (hits_host) memcpy(hits_host, c->hits.data(), nbin * sizeof(uint32_t));
    if (ndat_total) *ndat_total = c->ndat_total;
    if (integration_length) *integration_length = c->length;
  }
  return DSPSR_AMD_OK;
}

extern "C" const float* dspsr_amd_reduce_profiles_result(const dspsr_amd_comm* c, uint64_t* nfloat)
{
  if (!c || c->pending || c->rank != c->root) { if (nfloat) *nfloat = 0; return nullptr; }
  if (nfloat) *nfloat = c->mode == DSPSR_AMD_REDUCE_SUM ? c->n : (uint64_t)c->nranks * c->n;
  return (const float*)c->host;
}
