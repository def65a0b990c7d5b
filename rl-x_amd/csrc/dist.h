// dist.h -- data-parallel plumbing of the PPO update (dist.hip): RCCL communicator owned by the context,
// rank-local row compaction of the global minibatch permutation, batched advantage statistics.
#pragma once
#include "common.h"

namespace rlx {

// does this context take part in collectives (world > 1, or a test hook stands in for them)?
bool dist_active(const rlx_ctx* ctx);
// sum buf[n] (dtype 0: float32, 1: float64) over the ranks IN PLACE, ordered after the work queued on `producer` and
// before whatever is queued on it next.  No-op for a single rank without hook.
int dist_allreduce(rlx_ctx* ctx, void* buf, int64_t n, int dtype, hipStream_t producer);
// rows a rank-local minibatch is padded to (fixed launch shapes; rows beyond the actual count carry zero weight)
int dist_row_capacity(int mb_global, int n_local, int n_global);
// lidx[u][0 .. counts[u]) = LOCAL flattened indices (t * N_local + n - env_off) of the rows of global minibatch u that
// live on this rank, in the order of the global permutation; row stride `cap`
int dist_compact(rlx_ctx* ctx, const int32_t* perm, int n_upd, int mb_global, int n_local, int n_global, int env_off, int cap,
                 int32_t* lidx, int32_t* counts, int32_t* overflow, int32_t* dropped, hipStream_t st);
// stats[u] = {sum adv, sum adv^2, count, 0} over rows idx[u * stride + (0 .. count)) of every minibatch u (fp64, fixed order);
// count = counts[u] (device, ragged local minibatches) or fixed_count when counts == nullptr
// dropped (optional, [n_upd]): rows this rank's compaction had to drop per minibatch -> slot 3 of the record
int dist_adv_sums(const float* adv, const int32_t* idx, const int32_t* counts, int n_upd, int stride, int fixed_count,
                  double* stats, hipStream_t st, const int32_t* dropped = nullptr);
// after the statistics all-reduce: accumulate the global number of dropped rows (slot 3 of every record) into the
// context's overflow word 1
int dist_sum_dropped(rlx_ctx* ctx, const double* stats, int n_upd, hipStream_t st);
int32_t* dist_overflow_slot(rlx_ctx* ctx);
// metrics [n_upd, 10]: keep on every rank the partial sums, on rank 0 only the replicated values, so that one
// all-reduce(sum) yields the global metrics
int dist_mask_metrics(float* metrics, int n_upd, int rank, int discrete, hipStream_t st);

}  // namespace rlx
