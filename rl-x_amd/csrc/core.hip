// core.hip -- context, error state, scratch arenas.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

namespace rlx {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

void* scratch(rlx_ctx* ctx, ScratchSlot s, size_t bytes) {
  Scratch& sl = ctx->slots[ctx->bank][s];
  if (sl.bytes >= bytes && sl.ptr) return sl.ptr;
  if (sl.ptr) {
    // growing: previous users of this slot may still be in flight
    if (hipDeviceSynchronize() != hipSuccess) { set_error("hipDeviceSynchronize failed in scratch()"); return nullptr; }
    (void)hipFree(sl.ptr);
    sl.ptr = nullptr;
    sl.bytes = 0;
  }
  // large arenas get 1/8 of headroom: the sharded update's local minibatches differ by a few rows from one permutation to the
  // next, and every growth costs a device synchronisation (288 GB of HBM: the slack is free)
  size_t want = ((bytes > (size_t(1) << 20) ? bytes + bytes / 8 : bytes) + 255) & ~size_t(255);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    set_error(std::string("hipMalloc(") + std::to_string(want) + ") failed: " + hipGetErrorString(e));
    return nullptr;
  }
  sl.ptr = p;
  sl.bytes = want;
  return p;
}

const float* zeros_f32(rlx_ctx* ctx, size_t n) {
  Scratch& sl = ctx->slots[0][SL_ZEROS];
  if (sl.ptr && sl.bytes >= n * sizeof(float)) return (const float*)sl.ptr;
  const int bank = ctx->bank;
  ctx->bank = 0;
  void* p = scratch(ctx, SL_ZEROS, n * sizeof(float));
  ctx->bank = bank;
  if (!p) return nullptr;
  if (hipMemset(p, 0, ctx->slots[0][SL_ZEROS].bytes) != hipSuccess) { set_error("hipMemset failed in zeros_f32()"); return nullptr; }
  return (const float*)p;
}

int ctx_side_stream(rlx_ctx* ctx) {
  if (ctx->side) return RLX_OK;
  RLX_HIP_TRY(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  for (int p = 0; p < 2; ++p) {
    RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_rows[p], hipEventDisableTiming));
    RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_cdone[p], hipEventDisableTiming));
  }
  return RLX_OK;
}

int ctx_sac_streams(rlx_ctx* ctx) {
  int rc = ctx_side_stream(ctx);
  if (rc) return rc;
  for (int i = 0; i < 2; ++i)
    if (!ctx->sac_st[i]) RLX_HIP_TRY(hipStreamCreateWithFlags(&ctx->sac_st[i], hipStreamNonBlocking));
  for (int i = 0; i < 6; ++i)
    if (!ctx->sac_ev[i]) RLX_HIP_TRY(hipEventCreateWithFlags(&ctx->sac_ev[i], hipEventDisableTiming));
  return RLX_OK;
}

static hipEvent_t prof_event(rlx_ctx* ctx) {
  if (!ctx->prof_pool.empty()) {
    hipEvent_t e = ctx->prof_pool.back();
    ctx->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(rlx_ctx* c, int kid, double flops, hipStream_t s, double bytes, int64_t M, int N, int K, int engine)
    : ctx(c), st(s) {
  if (!c || !c->prof_on) return;
  int row = -1;
  for (size_t i = 0; i < c->prof_rows.size(); ++i) {
    const ProfRow& q = c->prof_rows[i];
    if (q.kid == kid && q.engine == engine && q.M == M && q.N == N && q.K == K) { row = (int)i; break; }
  }
  if (row < 0) {
    row = (int)c->prof_rows.size();
    c->prof_rows.push_back(ProfRow{kid, engine, M, N, K, 0, 0, 0.0, 0.0, 0.0});
  }
  const int64_t seq = c->prof_rows[row].launches++;
  if (c->prof_sample > 1 && (seq % c->prof_sample) != 0) return;
  ProfRec r{kid, row, flops, bytes, prof_event(c), prof_event(c)};
  idx = (int)c->prof_recs.size();
  c->prof_recs.push_back(r);
}

// max |x| of a float array as the bit pattern of a non-negative float (ordered like the unsigned integers): per-block maximum,
// then ONE atomicMax per block on a word a preceding memset zeroed (order-independent: deterministic)
__global__ __launch_bounds__(256) void k_abs_max(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  __shared__ float s_m[4];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && (reinterpret_cast<uintptr_t>(x + i) & 15) == 0) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    } else {
      for (int j = 0; j < 4 && i + j < n; ++j) m = fmaxf(m, fabsf(x[i + j]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));      // (+inf included; a NaN never wins an fmaxf)
  }
}

int x_max_update(rlx_ctx* ctx, const float* x, int64_t n, int which, hipStream_t st, const uint32_t** out) {
  const int bank = ctx->bank;
  ctx->bank = 0;
  uint32_t* slot = (uint32_t*)scratch(ctx, SL_XMAX, 4 * sizeof(uint32_t));
  ctx->bank = bank;
  if (!slot) return RLX_ENOMEM;
  slot += which & 3;
  RLX_HIP_TRY(hipMemsetAsync(slot, 0, sizeof(uint32_t), st));
  int grid = div_up(n, 256 * 4 * 8);
  grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);
  hipLaunchKernelGGL(k_abs_max, dim3(grid), dim3(256), 0, st, x, n, slot);
  RLX_LAUNCH_CHECK();
  *out = slot;
  return RLX_OK;
}

hipEvent_t ProfScope::ev0() const { return idx >= 0 ? ctx->prof_recs[idx].e0 : nullptr; }
hipEvent_t ProfScope::ev1() const { return idx >= 0 ? ctx->prof_recs[idx].e1 : nullptr; }

}  // namespace rlx

extern "C" {

int rlx_prof_begin(rlx_ctx* ctx) {
  RLX_REQUIRE(ctx, RLX_EINVAL, "rlx_prof_begin: ctx is NULL");
  if (!ctx->prof_ref) RLX_HIP_TRY(hipEventCreate(&ctx->prof_ref));
  RLX_HIP_TRY(hipDeviceSynchronize());
  RLX_HIP_TRY(hipEventRecord(ctx->prof_ref, 0));
  ctx->prof_rows.clear();
  ctx->prof_on = true;
  return RLX_OK;
}

int rlx_prof_kernel_count(void) { return rlx::PK_COUNT; }

const char* rlx_prof_kernel_name(int k) {
  static const char* names[rlx::PK_COUNT] = {"k_gemm_fwd", "k_gemm_dx", "k_gemm_dw", "k_dx_l1bwd",
                                              "k_l1fwd_mfma", "k_head_loss", "k_reduce_segments", "k_l12fwd", "k_tail", "k_fwd2h"};
  return (k >= 0 && k < rlx::PK_COUNT) ? names[k] : nullptr;
}

int rlx_prof_end(rlx_ctx* ctx, double* ms_out, double* flops_out, double* bytes_out, int64_t* count_out) {
  RLX_REQUIRE(ctx && ms_out && flops_out && bytes_out && count_out, RLX_EINVAL, "rlx_prof_end: NULL pointer");
  ctx->prof_on = false;
  RLX_HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < rlx::PK_COUNT; ++k) { ms_out[k] = 0.0; flops_out[k] = 0.0; bytes_out[k] = 0.0; count_out[k] = 0; }
  std::vector<std::pair<float, float>> iv;   // [start, end) of every launch relative to prof_ref (all streams)
  iv.reserve(ctx->prof_recs.size());
  for (auto& r : ctx->prof_recs) {
    float ms = 0.f, t0 = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      ms_out[r.kid] += ms;
      flops_out[r.kid] += r.flops;
      bytes_out[r.kid] += r.bytes;
      count_out[r.kid] += 1;
      rlx::ProfRow& q = ctx->prof_rows[r.row];
      q.timed += 1;
      q.ms += ms;
      q.flops += r.flops;
      q.bytes += r.bytes;
      // (the union of intervals is the MATRIX kernels' busy time: memory-bound rows stay out of it)
      if (q.engine != rlx::PROF_ENGINE_HBM && ctx->prof_ref && hipEventElapsedTime(&t0, ctx->prof_ref, r.e0) == hipSuccess)
        iv.emplace_back(t0, t0 + ms);
    }
    ctx->prof_pool.push_back(r.e0);
    ctx->prof_pool.push_back(r.e1);
  }
  ctx->prof_recs.clear();
  std::sort(iv.begin(), iv.end());
  double uni = 0.0;
  float cs = 0.f, ce = -1.f;
  for (auto& p : iv) {
    if (ce < 0.f || p.first > ce) { if (ce >= 0.f) uni += ce - cs; cs = p.first; ce = p.second; }
    else if (p.second > ce) ce = p.second;
  }
  if (ce >= 0.f) uni += ce - cs;
  ctx->prof_union_ms = uni;
  return RLX_OK;
}

int rlx_prof_rows(rlx_ctx* ctx, rlx_prof_row* rows, int capacity, int* n_out) {
  RLX_REQUIRE(ctx && n_out && (rows || capacity == 0), RLX_EINVAL, "rlx_prof_rows: NULL pointer");
  const int n = (int)ctx->prof_rows.size();
  *n_out = n;
  for (int i = 0; i < n && i < capacity; ++i) {
    const rlx::ProfRow& q = ctx->prof_rows[i];
    rows[i] = rlx_prof_row{q.kid, q.engine, q.N, q.K, q.M, q.launches, q.timed, q.ms, q.flops, q.bytes};
  }
  return RLX_OK;
}

int rlx_prof_union_ms(rlx_ctx* ctx, double* out) {
  RLX_REQUIRE(ctx && out, RLX_EINVAL, "rlx_prof_union_ms: NULL pointer");
  *out = ctx->prof_union_ms;
  return RLX_OK;
}

int rlx_dbg_set_option(rlx_ctx* ctx, const char* name, int value) {
  RLX_REQUIRE(ctx && name, RLX_EINVAL, "rlx_dbg_set_option: NULL");
  if (std::string(name) == "sac_twin") { ctx->sac_twin = value; return RLX_OK; }
  if (std::string(name) == "disable_l1fused") { ctx->disable_l1fused = value != 0; return RLX_OK; }
  if (std::string(name) == "l1fwd_mfma") { ctx->l1fwd_mfma = value != 0; return RLX_OK; }
  if (std::string(name) == "pipeline_updates") { ctx->pipeline_updates = value != 0; return RLX_OK; }
  if (std::string(name) == "two_streams") { ctx->two_streams = value != 0; return RLX_OK; }
  if (std::string(name) == "ppo_twin") { ctx->ppo_twin = value; return RLX_OK; }
  if (std::string(name) == "ro_exit") { ctx->ro_exit = value; return RLX_OK; }
  if (std::string(name) == "lf_idle_cus") { ctx->lf_idle_cus = value < 0 ? 0 : value; return RLX_OK; }
  if (std::string(name) == "dw_recompute") { ctx->dw_recompute = value != 0; return RLX_OK; }
  if (std::string(name) == "dw_merge") { ctx->dw_merge = value != 0; return RLX_OK; }
  if (std::string(name) == "fwd2h") { ctx->fwd2h = value != 0; return RLX_OK; }
  if (std::string(name) == "l12_fused") { ctx->l12_fused = value != 0; return RLX_OK; }
  if (std::string(name) == "gather_group_rows") { ctx->gather_group_rows = value; return RLX_OK; }
  if (std::string(name) == "gather_records") { ctx->gather_records = value != 0; return RLX_OK; }
  if (std::string(name) == "ppo_tail") { ctx->ppo_tail = value < 0 ? -1 : (value > 2 ? 2 : value); return RLX_OK; }
  if (std::string(name) == "bx_force_mi") { ctx->bx_force_mi = value; return RLX_OK; }
  if (std::string(name) == "adam_emit") { ctx->adam_emit = value != 0; return RLX_OK; }
  if (std::string(name) == "bx_debug") { ctx->bx_debug = value; return RLX_OK; }
  if (std::string(name) == "bx_gscale_log2") { ctx->bx_gscale = ldexpf(1.f, value < 0 ? 0 : (value > 40 ? 40 : value)); return RLX_OK; }
  if (std::string(name) == "gemm_bx") { ctx->gemm_bx = value != 0; return RLX_OK; }
  if (std::string(name) == "prof_sample") { ctx->prof_sample = value < 1 ? 1 : value; return RLX_OK; }
  if (std::string(name) == "fused_recurrent_act") { ctx->fused_recurrent_act = value != 0; return RLX_OK; }
  RLX_REQUIRE(false, RLX_EINVAL, "rlx_dbg_set_option: unknown option");
}

int rlx_dbg_get_counter(rlx_ctx* ctx, const char* name, int64_t* out) {
  RLX_REQUIRE(ctx && name && out, RLX_EINVAL, "rlx_dbg_get_counter: NULL");
  int bank = 0, slot = 0;
  if (sscanf(name, "scratch_ptr:%d:%d", &bank, &slot) == 2 && bank >= 0 && bank < 3 && slot >= 0 && slot < rlx::SL_COUNT) {
    *out = (int64_t)reinterpret_cast<uintptr_t>(ctx->slots[bank][slot].ptr);
    return RLX_OK;
  }
  if (sscanf(name, "scratch_bytes:%d:%d", &bank, &slot) == 2 && bank >= 0 && bank < 3 && slot >= 0 && slot < rlx::SL_COUNT) {
    *out = (int64_t)ctx->slots[bank][slot].bytes;
    return RLX_OK;
  }
  if (std::string(name) == "allreduce_calls") { *out = ctx->ar_calls; return RLX_OK; }
  if (std::string(name) == "bx_window_fallbacks") { *out = ctx->bx_window_fallbacks; return RLX_OK; }
  if (std::string(name) == "gemm_bx") { *out = ctx->gemm_bx ? 1 : 0; return RLX_OK; }
  RLX_REQUIRE(false, RLX_EINVAL, "rlx_dbg_get_counter: unknown counter");
}

int rlx_version(void) { return 200; }

const char* rlx_last_error(void) { return rlx::g_last_error.c_str(); }

int rlx_ctx_create(int device, rlx_ctx** out) {
  RLX_REQUIRE(out != nullptr, RLX_EINVAL, "rlx_ctx_create: out is NULL");
  int count = 0;
  RLX_HIP_TRY(hipGetDeviceCount(&count));
  RLX_REQUIRE(device >= 0 && device < count, RLX_EINVAL, "rlx_ctx_create: no such HIP device");
  RLX_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  RLX_HIP_TRY(hipGetDeviceProperties(&prop, device));
  rlx_ctx* c = new rlx_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  if (const char* e = getenv("RLX_GEMM_BX")) c->gemm_bx = atoi(e) != 0;   // engine A/B without touching the caller
  if (const char* e = getenv("RLX_BX_DEBUG")) c->bx_debug = atoi(e);       // which kernel classes stay on the exact engine (option "bx_debug")
  *out = c;
  return RLX_OK;
}

int rlx_dist_release(rlx_ctx* ctx);   // dist.hip

int rlx_ctx_destroy(rlx_ctx* ctx) {
  if (!ctx) return RLX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  (void)rlx_dist_release(ctx);
  for (int i = 0; i < 2; ++i)
    if (ctx->sac_st[i]) (void)hipStreamDestroy(ctx->sac_st[i]);
  for (int i = 0; i < 6; ++i)
    if (ctx->sac_ev[i]) (void)hipEventDestroy(ctx->sac_ev[i]);
  for (int i = 0; i < 4; ++i) {
    if (ctx->sched_host[i]) (void)hipHostFree(ctx->sched_host[i]);
    if (ctx->sched_ev[i]) (void)hipEventDestroy(ctx->sched_ev[i]);
  }
  for (int b = 0; b < 3; ++b)
    for (int i = 0; i < rlx::SL_COUNT; ++i)
      if (ctx->slots[b][i].ptr) (void)hipFree(ctx->slots[b][i].ptr);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (int p = 0; p < 2; ++p) {
    if (ctx->ev_rows[p]) (void)hipEventDestroy(ctx->ev_rows[p]);
    if (ctx->ev_cdone[p]) (void)hipEventDestroy(ctx->ev_cdone[p]);
  }
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
  if (ctx->prof_ref) (void)hipEventDestroy(ctx->prof_ref);
  if (ctx->pf_done) (void)hipEventDestroy(ctx->pf_done);
  if (ctx->ev_perm_free) (void)hipEventDestroy(ctx->ev_perm_free);
  delete ctx;
  return RLX_OK;
}

int64_t rlx_mlp_param_count(const rlx_mlp_desc* desc) {
  if (!desc || desc->n_hidden < 1 || desc->n_hidden > 3) return -1;
  return rlx::make_layout(*desc).n_params;
}

}  // extern "C"
