// ppo_internal.h -- pieces of ppo.hip reused by the recurrent variant (ppo_lstm.hip).
#pragma once
#include "mlp.h"

namespace rlx {

struct MbScratch {
  float* mb_x;     // [mb, O]  gathered observations
  float* mb_xc = nullptr;   // [mb, Oc] gathered CRITIC observations when the critic reads its own observation columns
                            // (rlx_ppo_hparams.critic_states; nullptr: the critic reads mb_x like the policy)
  float* mb_a;     // [mb, A]  gathered actions
  float* aux;      // [mb, 3]  log_prob, return, advantage
  double* stats;   // {sum adv, sum adv^2, count}
  float* acts[4] = {nullptr, nullptr, nullptr, nullptr};  // activation buffers of the MLP nets ([3]: pre-LayerNorm values of a wide first layer)
  float* head_part;
  const int32_t* valid_rows = nullptr;   // device, optional: rows [*valid_rows, mb) are zero-weight padding (data-parallel update)
  // optional source of the gather (ppo.hip: pack_rollout_rows): row i of the rollout as ONE aligned record of 4 << rec_lg4 floats
  // [obs | action | log_prob, return, advantage | pad] -- two cache lines per sampled row instead of six
  const float* rec = nullptr;
  int rec_lg4 = 0;
};

// gather rows idx[mb] of the flattened rollout arrays + fp64 advantage sums (K5)
// (cstates / Oc: the critic's own observation rows, see MbScratch.mb_xc; nullptr / 0 otherwise)
int ppo_gather(rlx_ctx* ctx, const float* states, const float* actions, const float* log_probs, const float* returns,
               const float* advantages, const int32_t* idx, int64_t mb, int O, int A, const MbScratch& s, hipStream_t st,
               const float* cstates = nullptr, int Oc = 0, bool local_stats = true);   // local_stats false: s.stats is already filled (all-reduced sums)
// policy output layer + PPO loss + seeds; h_last [mb,K] becomes dZ_last in place; head/logstd gradients reduced at once
int ppo_policy_head_loss(rlx_ctx* ctx, float* h_last, const float* Wh, const float* bh, const float* logstd,
                         const MbScratch& s, float* metrics, int64_t mb, int mb_global, int K, int A, int act,
                         const rlx_ppo_hparams& hp, float* gW, float* gb, float* glogstd, float* sumsq, int* nsq,
                         hipStream_t st);
// critic MLP: forward + value loss + backward into cgrads (metrics[1])
int ppo_critic_fwd_bwd(rlx_ctx* ctx, const rlx_mlp_desc& cd, const float* cparams, float* cgrads, float* metrics,
                       const MbScratch& s, int64_t mb, int mb_global, const rlx_ppo_hparams& hp, float* sumsq, int* nsq,
                       hipStream_t st);
// deterministic: a = mean (no noise);
// a = mean + exp(logstd) * normal(key, [N_global, A])[row_off + n], log-prob, optional clip/rescale and states_row copy
int ppo_sample(const float* mean, const float* logstd, uint32_t k0, uint32_t k1, int scheme, float* action, float* processed,
               float* logp, const float* obs, float* states_row, int N, int A, int O, int clip_and_rescale, const float* lo,
               const float* hi, int row_off, int N_global, hipStream_t st, int deterministic = 0);
// tail of the recurrent acting step in ONE launch (rollout.hip): policy torso on x [N, K0] + head + sampling, and the
// feed-forward critic on obs.  Offsets are relative to `params` (the flat recurrent-policy vector).
struct RolloutDecoder {
  const float* params;
  const float* x;       // [N, K0] torso input ([obs latent | cell latent]) -- or, with xb, only its first K0 - 64 columns
  const float* xb;      // optional [N, 64]: raw cell output; the kernel appends act(LayerNorm(xb)) (scale / bias at xb_g / xb_be)
  int64_t xb_g, xb_be;
  int K0, hidden[3], out_dim, act;
  int64_t W[3], b[3], g0, be0, headW, headb, logstd;
};
bool rollout_decoder_supported(const RolloutDecoder& p, const rlx_mlp_desc& cd);
int launch_rollout_decoder(rlx_ctx* ctx, const RolloutDecoder& p, const rlx_mlp_desc& cd, const float* cparams,
                           const float* obs, int O, uint32_t k0, uint32_t k1, int scheme, float* action, float* processed,
                           float* value, float* logp, int N, int clip_and_rescale, const float* lo, const float* hi,
                           int noise_row_offset, int N_global, int deterministic, hipStream_t st);
// scratch for a minibatch of mb rows (acts sized for `cd`; head partials for a policy head [Kp, A])
int ppo_mb_scratch(rlx_ctx* ctx, int O, int A, const rlx_mlp_desc& cd, int Kp, int64_t mb, MbScratch* s, bool critic_rows = false);

}  // namespace rlx
