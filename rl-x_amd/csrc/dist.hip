// dist.hip -- the data-parallel half of the PPO update on gfx950: RCCL communicator owned by the context, and the
// index / statistics kernels that turn the replicated GLOBAL minibatch permutation into rank-local work.
//
// The reference has no multi-device path; the contract is BASELINE.json's north_star + SURVEY.md 8(e):
// envs (and with them every [T, N, .] rollout array) are sharded over one process per GPU, parameters / Adam moments /
// the PRNG key are replicated, the permutation of rl_x/algorithms/ppo/flax/ppo.py:191-194 is computed identically on
// every rank over the global index space i = t * N_global + n (:180-184), and the only exchange step of the path is an
// all-reduce(sum) of the flat gradient vectors before clip + Adam (ppo.py:212-213) -- plus one all-reduce of the
// per-minibatch advantage sums (the normalisation of ppo.py:199-200 is over the GLOBAL minibatch) and one of the metric
// partial sums per iteration.
//
// RCCL is resolved at run time (dlsym / dlopen) instead of at link time: the process must hold ONE RCCL -- the one
// PyTorch-ROCm already mapped, which is bound to the same HIP runtime as the tensors we are handed -- and a link-time
// dependency on /opt/rocm/lib/librccl.so would map a second copy next to torch's bundled one.
#include "dist.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace rlx {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional (diagnostics: rlx_dist_comm_count)
  bool ok = false;
};
static RcclApi g_rccl;

static bool rccl_resolve(void* h) {
  RcclApi a;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.GetErrorString;
  if (a.ok) g_rccl = a;
  return a.ok;
}

static std::string g_rccl_path;   // what the entry points were bound to (rlx_dist_rccl_path)

// path of an RCCL this process has ALREADY mapped (PyTorch-ROCm maps its bundled copy when torch is imported), or ""
static std::string rccl_mapped_path() {
  std::string found;
  if (FILE* f = fopen("/proc/self/maps", "r")) {
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
      const char* p = strstr(line, "librccl");
      if (!p) continue;
      const char* s = strchr(line, '/');
      if (!s) continue;
      std::string path(s);
      while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
      found = path;
      break;
    }
    fclose(f);
  }
  return found;
}

// ONE RCCL per process.  Order: symbols already visible globally; the copy the process has mapped (handle without loading:
// RTLD_NOLOAD); only when NO librccl is mapped at all a fresh copy from the candidates.  A second copy next to a mapped one
// (another HIP runtime binding, its own proxy threads) is never loaded: that case fails instead.
static int rccl_load(const char* path) {
  if (g_rccl.ok) return RLX_OK;
  if (rccl_resolve(RTLD_DEFAULT)) { g_rccl_path = "(global symbols)"; return RLX_OK; }
  const std::string mapped = rccl_mapped_path();
  if (!mapped.empty()) {
    void* h = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (h && rccl_resolve(h)) { g_rccl_path = mapped; return RLX_OK; }
    set_error("an RCCL is mapped into this process (" + mapped + ") but its entry points cannot be bound; refusing to load a second copy");
    return RLX_EUNSUP;
  }
  const char* cands[] = {path, getenv("RLX_RCCL_LIBRARY"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* c : cands) {
    if (!c || !*c) continue;
    void* h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h && rccl_resolve(h)) { g_rccl_path = c; return RLX_OK; }
  }
  set_error("RCCL not found: pass the path of librccl.so (torch/lib/librccl.so) to rlx_dist_load_rccl or set RLX_RCCL_LIBRARY");
  return RLX_EUNSUP;
}

#define RLX_NCCL_TRY(expr)                                                                                      \
  do {                                                                                                          \
    ncclResult_t r_ = (expr);                                                                                   \
    if (r_ != ncclSuccess) {                                                                                    \
      ::rlx::set_error(std::string(#expr) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?")); \
      return RLX_EHIP;                                                                                          \
    }                                                                                                           \
  } while (0)

bool dist_active(const rlx_ctx* ctx) { return ctx->world > 1 || ctx->comm != nullptr || ctx->ar_hook != nullptr; }

int dist_allreduce(rlx_ctx* ctx, void* buf, int64_t n, int dtype, hipStream_t producer) {
  if (ctx->ar_hook || ctx->world > 1 || ctx->comm) ++ctx->ar_calls;   // collectives actually issued (rlx_dbg_get_counter "allreduce_calls")
  if (ctx->ar_hook) {
    const int rc = ctx->ar_hook(ctx->ar_hook_user, buf, n, dtype, producer == ctx->side && ctx->side ? 1 : 0);
    RLX_REQUIRE(rc == 0, RLX_EINVAL, "all-reduce hook failed");
    return RLX_OK;
  }
  if (ctx->world <= 1 && !ctx->comm) return RLX_OK;
  RLX_REQUIRE(ctx->comm && ctx->comm_stream, RLX_EINVAL, "data-parallel context without a communicator");
  // one communicator, one stream: the collectives of both chains (policy on the caller's stream, critic on the side
  // stream) are enqueued in program order, which is the same on every rank -- no cross-communicator ordering hazards
  hipEvent_t ready = ctx->comm_ev[ctx->comm_ev_pos], done = ctx->comm_ev[ctx->comm_ev_pos + 1];
  ctx->comm_ev_pos = (ctx->comm_ev_pos + 2) & 31;
  RLX_HIP_TRY(hipEventRecord(ready, producer));
  RLX_HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ready, 0));
  RLX_NCCL_TRY(g_rccl.AllReduce(buf, buf, (size_t)n, dtype ? ncclDouble : ncclFloat, ncclSum, (ncclComm_t)ctx->comm,
                                ctx->comm_stream));
  RLX_HIP_TRY(hipEventRecord(done, ctx->comm_stream));
  RLX_HIP_TRY(hipStreamWaitEvent(producer, done, 0));
  return RLX_OK;
}

int dist_row_capacity(int mb_global, int n_local, int n_global) {
  if (n_local >= n_global) return mb_global;
  // local rows of a global minibatch ~ Binomial(mb, p) at worst (sampling without replacement is narrower):
  // mean + 6.5 sigma + slack, rounded to whole 128-row GEMM tiles; P(overflow) < 1e-10 per minibatch, detected on the device
  const double p = (double)n_local / (double)n_global;
  const double mean = mb_global * p, sigma = sqrt(mb_global * p * (1.0 - p));
  int64_t cap = (int64_t)ceil(mean + 6.5 * sigma + 64.0);
  cap = ((cap + 127) / 128) * 128;
  return (int)(cap < mb_global ? cap : mb_global);
}

// ---------------------------------------------------------------------------------------
// Stable per-minibatch compaction: one workgroup per global minibatch walks its mb permutation entries in chunks of
// 1024 (4 per thread), keeps the rows whose env lives on this rank and writes their LOCAL flattened index at the running
// offset (block-wide exclusive scan of the keep counts: in-thread, then DPP-free wave scan, then 4 wave totals in LDS).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_local(const int32_t* __restrict__ perm, int32_t* __restrict__ lidx,
                                                       int32_t* __restrict__ counts, int32_t* __restrict__ overflow,
                                                       int32_t* __restrict__ dropped, int mb, int n_global, int n_local,
                                                       int env_off, int cap) {
  __shared__ int s_wave[4];
  const int u = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int32_t* src = perm + (int64_t)u * mb;
  int32_t* dst = lidx + (int64_t)u * cap;
  int base = 0;
  for (int c0 = 0; c0 < mb; c0 += 1024) {
    int loc[4];
    bool keep[4];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = c0 + 4 * t + j;
      keep[j] = false;
      loc[j] = 0;
      if (i < mb) {
        const int p = src[i];
        const int tt = p / n_global, n = p - tt * n_global;
        keep[j] = n >= env_off && n < env_off + n_local;
        loc[j] = tt * n_local + (n - env_off);
      }
      cnt += keep[j] ? 1 : 0;
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < w) wave_off += s_wave[q];
      total += s_wave[q];
    }
    int pos = base + wave_off + incl - cnt;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (keep[j]) {
        if (pos < cap) dst[pos] = loc[j];
        ++pos;
      }
    base += total;
    __syncthreads();
  }
  if (t == 0) {
    // rows beyond the padded capacity are dropped; their number rides in slot 3 of this minibatch's statistics record, which
    // the update all-reduces anyway -- every rank then learns of the overflow in the same iteration (rlx_dist_overflow_count)
    if (dropped) dropped[u] = base > cap ? base - cap : 0;
    if (base > cap) {
      atomicAdd(overflow, 1);
      base = cap;
    }
    counts[u] = base;
  }
}

// stats[u] = {sum adv, sum adv^2, count, 0} over the rows idx[u * stride + (0 .. count)) of minibatch u; count = counts[u]
// (rank-local, ragged) or fixed_count.  One workgroup per minibatch, fp64, fixed summation order: thread t owns rows
// t, t + 256, ..., then a butterfly over the wave and the four waves in order -- reproducible bit for bit.
__global__ __launch_bounds__(256) void k_mb_adv_sums(const float* __restrict__ adv, const int32_t* __restrict__ lidx,
                                                     const int32_t* __restrict__ counts, int stride, int fixed_count,
                                                     double* __restrict__ stats, const int32_t* __restrict__ dropped) {
  __shared__ double s_red[8];
  const int u = blockIdx.x, cnt = counts ? counts[u] : fixed_count;
  const int32_t* idx = lidx + (int64_t)u * stride;
  double s1 = 0.0, s2 = 0.0;
  int r = threadIdx.x;
  for (; r + 768 < cnt; r += 1024) {   // four independent gathers in flight
    const float a0 = adv[idx[r]], a1 = adv[idx[r + 256]], a2 = adv[idx[r + 512]], a3 = adv[idx[r + 768]];
    s1 += (double)a0; s2 += (double)a0 * (double)a0;
    s1 += (double)a1; s2 += (double)a1 * (double)a1;
    s1 += (double)a2; s2 += (double)a2 * (double)a2;
    s1 += (double)a3; s2 += (double)a3 * (double)a3;
  }
  for (; r < cnt; r += 256) {
    const double a = (double)adv[idx[r]];
    s1 += a;
    s2 += a * a;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[w] = s1; s_red[4 + w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[4 * u + 0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    stats[4 * u + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    stats[4 * u + 2] = (double)cnt;
    stats[4 * u + 3] = dropped ? (double)dropped[u] : 0.0;   // rows this rank had to drop (capacity overflow; see k_compact_local)
  }
}

// after the statistics all-reduce: slot 3 of every record holds the GLOBAL number of dropped rows of that minibatch -- the
// same value on every rank.  Accumulated into the context's second overflow word (read and reset by rlx_dist_overflow_count).
__global__ __launch_bounds__(256) void k_sum_dropped(const double* __restrict__ stats, int n_upd, int32_t* __restrict__ total) {
  __shared__ int s_red[4];
  int v = 0;
  for (int u = threadIdx.x; u < n_upd; u += 256) v += (int)stats[4 * u + 3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) total[0] += s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ void k_mask_metrics(float* __restrict__ met, int n, int rank, int discrete) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = i % 10;
  // 0 pg loss, 1 critic loss, 3 approx KL, 4 clip fraction (and 2 = entropy for a Categorical policy) are partial sums
  // over this rank's rows; 2 (Gaussian entropy), 5 / 6 advantage mean / std, 7 policy std, 8 / 9 gradient norms are
  // replicated values: rank 0 alone contributes them to the sum
  const bool partial = c == 0 || c == 1 || c == 3 || c == 4 || (c == 2 && discrete);
  if (!partial && rank != 0) met[i] = 0.f;
}

// device counter of capacity overflows (zeroed when it is first allocated)
int32_t* dist_overflow_slot(rlx_ctx* ctx) {
  const bool fresh = ctx->slots[0][SL_OVERFLOW].ptr == nullptr;
  const int bank = ctx->bank;
  ctx->bank = 0;
  int32_t* ovf = (int32_t*)scratch(ctx, SL_OVERFLOW, 64);
  ctx->bank = bank;
  if (ovf && fresh && hipMemset(ovf, 0, 64) != hipSuccess) { set_error("hipMemset failed in dist_overflow_slot()"); return nullptr; }
  return ovf;
}

int dist_compact(rlx_ctx* ctx, const int32_t* perm, int n_upd, int mb_global, int n_local, int n_global, int env_off, int cap,
                 int32_t* lidx, int32_t* counts, int32_t* overflow, int32_t* dropped, hipStream_t st) {
  hipLaunchKernelGGL(k_compact_local, dim3(n_upd), dim3(256), 0, st, perm, lidx, counts, overflow, dropped, mb_global,
                     n_global, n_local, env_off, cap);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int dist_adv_sums(const float* adv, const int32_t* idx, const int32_t* counts, int n_upd, int stride, int fixed_count,
                  double* stats, hipStream_t st, const int32_t* dropped) {
  hipLaunchKernelGGL(k_mb_adv_sums, dim3(n_upd), dim3(256), 0, st, adv, idx, counts, stride, fixed_count, stats, dropped);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int dist_sum_dropped(rlx_ctx* ctx, const double* stats, int n_upd, hipStream_t st) {
  int32_t* ovf = dist_overflow_slot(ctx);
  if (!ovf) return RLX_ENOMEM;
  hipLaunchKernelGGL(k_sum_dropped, dim3(1), dim3(256), 0, st, stats, n_upd, ovf + 1);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

int dist_mask_metrics(float* metrics, int n_upd, int rank, int discrete, hipStream_t st) {
  const int n = n_upd * 10;
  hipLaunchKernelGGL(k_mask_metrics, dim3(div_up(n, 256)), dim3(256), 0, st, metrics, n, rank, discrete);
  RLX_LAUNCH_CHECK();
  return RLX_OK;
}

}  // namespace rlx

using namespace rlx;

extern "C" {

int rlx_dist_load_rccl(const char* path) { return rccl_load(path); }

const char* rlx_dist_rccl_path(void) { return g_rccl_path.c_str(); }

int rlx_dist_unique_id(void* id_out) {
  RLX_REQUIRE(id_out, RLX_EINVAL, "rlx_dist_unique_id: NULL pointer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  ncclUniqueId id;
  RLX_NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return RLX_OK;
}

int rlx_ctx_create_dist(int device, int rank, int world, const void* nccl_unique_id, rlx_ctx** out) {
  RLX_REQUIRE(world >= 1 && rank >= 0 && rank < world, RLX_EINVAL, "rlx_ctx_create_dist: need 0 <= rank < world");
  RLX_REQUIRE(world == 1 || nccl_unique_id, RLX_EINVAL, "rlx_ctx_create_dist: world > 1 needs the 128-byte unique id of rank 0");
  rlx_ctx* ctx = nullptr;
  int rc = rlx_ctx_create(device, &ctx);
  if (rc) return rc;
  ctx->rank = rank;
  ctx->world = world;
  if (nccl_unique_id) {   // (a one-rank job may own a communicator too: the collectives are then really enqueued)
    rc = rccl_load(nullptr);
    if (rc) { rlx_ctx_destroy(ctx); return rc; }
    ncclUniqueId id;
    memcpy(&id, nccl_unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
      set_error(std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r));
      rlx_ctx_destroy(ctx);
      return RLX_EHIP;
    }
    ctx->comm = comm;
    if (hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking) != hipSuccess) {
      set_error("rlx_ctx_create_dist: hipStreamCreate failed");
      rlx_ctx_destroy(ctx);
      return RLX_EHIP;
    }
    for (int i = 0; i < 32; ++i)
      if (hipEventCreateWithFlags(&ctx->comm_ev[i], hipEventDisableTiming) != hipSuccess) {
        set_error("rlx_ctx_create_dist: hipEventCreate failed");
        rlx_ctx_destroy(ctx);
        return RLX_EHIP;
      }
  }
  *out = ctx;
  return RLX_OK;
}

// called by rlx_ctx_destroy (core.hip)
int rlx_dist_release(rlx_ctx* ctx) {
  if (!ctx) return RLX_OK;
  if (ctx->comm && g_rccl.ok) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  for (int i = 0; i < 32; ++i)
    if (ctx->comm_ev[i]) { (void)hipEventDestroy(ctx->comm_ev[i]); ctx->comm_ev[i] = nullptr; }
  if (ctx->comm_stream) { (void)hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
  return RLX_OK;
}

int rlx_ctx_rank(const rlx_ctx* ctx, int* rank, int* world) {
  RLX_REQUIRE(ctx && rank && world, RLX_EINVAL, "rlx_ctx_rank: NULL pointer");
  *rank = ctx->rank;
  *world = ctx->world;
  return RLX_OK;
}

int rlx_dist_comm_count(const rlx_ctx* ctx, int* count_out) {
  RLX_REQUIRE(ctx && count_out, RLX_EINVAL, "rlx_dist_comm_count: NULL pointer");
  *count_out = 0;
  if (!ctx->comm) return RLX_OK;                       // no communicator (single rank, or collectives through the test hook)
  RLX_REQUIRE(g_rccl.ok && g_rccl.CommCount, RLX_EUNSUP, "rlx_dist_comm_count: the bound RCCL has no ncclCommCount");
  RLX_NCCL_TRY(g_rccl.CommCount((ncclComm_t)ctx->comm, count_out));
  return RLX_OK;
}

int rlx_allreduce_grads(rlx_ctx* ctx, float* buf, int64_t n, void* stream) {
  RLX_REQUIRE(ctx && buf && n > 0, RLX_EINVAL, "rlx_allreduce_grads: bad args");
  ctx->sac_img.written(buf, n);
  return dist_allreduce(ctx, buf, n, 0, (hipStream_t)stream);
}

int rlx_dbg_set_allreduce_hook(rlx_ctx* ctx, rlx_allreduce_fn fn, void* user) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_dbg_set_allreduce_hook: ctx is NULL");
  ctx->ar_hook = fn;
  ctx->ar_hook_user = user;
  return RLX_OK;
}

int rlx_dbg_set_rank(rlx_ctx* ctx, int rank, int world) {
  RLX_REQUIRE(ctx && world >= 1 && rank >= 0 && rank < world, RLX_EINVAL, "rlx_dbg_set_rank: need 0 <= rank < world");
  RLX_REQUIRE(!ctx->comm, RLX_EINVAL, "rlx_dbg_set_rank: the context owns a communicator");
  ctx->rank = rank;
  ctx->world = world;
  return RLX_OK;
}

int rlx_dist_row_capacity(int mb_global, int n_local, int n_global) {
  if (mb_global <= 0 || n_local <= 0 || n_global < n_local) return -1;
  return dist_row_capacity(mb_global, n_local, n_global);
}

int rlx_dist_local_rows_i32(rlx_ctx* ctx, const int32_t* perm, int n_minibatches, int mb_global, int n_local, int n_global,
                            int env_id_offset, int cap, int32_t* lidx, int32_t* counts, void* stream) {
  RLX_REQUIRE(ctx && perm && lidx && counts && n_minibatches > 0 && mb_global > 0 && n_local > 0 && n_global >= n_local &&
                  env_id_offset >= 0 && env_id_offset + n_local <= n_global && cap > 0,
              RLX_EINVAL, "rlx_dist_local_rows_i32: bad args");
  int32_t* ovf = dist_overflow_slot(ctx);
  if (!ovf) return RLX_ENOMEM;
  return dist_compact(ctx, perm, n_minibatches, mb_global, n_local, n_global, env_id_offset, cap, lidx, counts, ovf, nullptr,
                      (hipStream_t)stream);
}

int rlx_dist_overflow_count(rlx_ctx* ctx, int* rows_any_rank, int* minibatches_this_rank, void* stream) {
  RLX_REQUIRE(ctx && rows_any_rank, RLX_EINVAL, "rlx_dist_overflow_count: NULL pointer");
  *rows_any_rank = 0;
  if (minibatches_this_rank) *minibatches_this_rank = 0;
  Scratch& sl = ctx->slots[0][SL_OVERFLOW];
  if (!sl.ptr) return RLX_OK;
  // word 0: minibatches THIS rank truncated (rlx_dist_local_rows_i32 and the update's own compaction) -- rank-local, a
  // diagnostic; word 1: rows dropped by ANY rank, summed from the all-reduced statistics records inside
  // rlx_ppo_update_dist_f32 -- identical on every rank, the figure a fatal-on-all-ranks decision has to be taken on.
  // Read and reset ON THE CALLER'S STREAM: ordered behind k_compact_local / k_sum_dropped of the updates issued on it.
  hipStream_t st = (hipStream_t)stream;
  int32_t v[2] = {0, 0};
  RLX_HIP_TRY(hipMemcpyAsync(v, sl.ptr, sizeof(v), hipMemcpyDeviceToHost, st));
  RLX_HIP_TRY(hipMemsetAsync(sl.ptr, 0, sizeof(v), st));
  RLX_HIP_TRY(hipStreamSynchronize(st));
  *rows_any_rank = v[1];
  if (minibatches_this_rank) *minibatches_this_rank = v[0];
  return RLX_OK;
}

}  // extern "C"
