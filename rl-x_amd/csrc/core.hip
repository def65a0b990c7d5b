// core.hip -- context, error state, scratch arenas.
#include "common.h"

namespace rlx {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

void* scratch(rlx_ctx* ctx, ScratchSlot s, size_t bytes) {
  Scratch& sl = ctx->slots[s];
  if (sl.bytes >= bytes && sl.ptr) return sl.ptr;
  if (sl.ptr) {
    // growing: previous users of this slot may still be in flight
    if (hipDeviceSynchronize() != hipSuccess) { set_error("hipDeviceSynchronize failed in scratch()"); return nullptr; }
    (void)hipFree(sl.ptr);
    sl.ptr = nullptr;
    sl.bytes = 0;
  }
  size_t want = (bytes + 255) & ~size_t(255);
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

}  // namespace rlx

extern "C" {

int rlx_version(void) { return 100; }

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
  *out = c;
  return RLX_OK;
}

int rlx_ctx_destroy(rlx_ctx* ctx) {
  if (!ctx) return RLX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < rlx::SL_COUNT; ++i)
    if (ctx->slots[i].ptr) (void)hipFree(ctx->slots[i].ptr);
  delete ctx;
  return RLX_OK;
}

int64_t rlx_mlp_param_count(const rlx_mlp_desc* desc) {
  if (!desc || desc->n_hidden < 1 || desc->n_hidden > 3) return -1;
  return rlx::make_layout(*desc).n_params;
}

}  // extern "C"
