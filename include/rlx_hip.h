/*
 * rlx_hip.h -- C ABI of librlxhip.so: the MI355X (gfx950) PPO/SAC training hot path
 * behind RL-X's plugin API.
 *
 * The reference (nico-bohlinger/RL-X) is 100 % Python; its "kernels" are XLA fusions
 * produced by jax.jit.  There is no FFI in the reference; each entry point below
 * names the reference jit region / function it replaces (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes stub a maintainer
 * would add on the reference side.
 *
 * Conventions
 *   - extern "C"; every function returns int: 0 = ok, negative = RLX_E*;
 *     rlx_last_error() returns a thread-local message valid until the next call.
 *   - tensor arguments are raw DEVICE pointers (contiguous, row-major, fp32 /
 *     int32 / uint32), caller-owned; the library never frees or retains them.
 *   - PRNG keys (uint32[2]) are HOST pointers (two words; split on the host).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     Launches are asynchronous; no implicit synchronisation.
 *   - library-owned scratch lives in the opaque rlx_ctx; one ctx per thread/GPU.
 *   - rollout arrays are time-major [T,N,...] exactly like the reference's Batch
 *     (rl_x/algorithms/ppo/flax/batch.py:1-11); flattened sample index i = t*N + n
 *     (rl_x/algorithms/ppo/flax/ppo.py:180-184).
 *
 * Flat parameter layout of one MLP (rlx_mlp_desc), shared with oracle/nets.py:
 *   for each hidden layer l: W_l[in,out] row-major (flax Dense kernel), b_l[out],
 *   then for layer 0 iff ln_first: ln_scale[out], ln_bias[out];
 *   head W[in,out], b[out]; iff has_logstd: logstd[out_dim].
 */
#ifndef RLX_HIP_H
#define RLX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLX_OK 0
#define RLX_EINVAL (-1)  /* bad argument                                  */
#define RLX_EHIP (-2)    /* a HIP runtime call failed (see rlx_last_error) */
#define RLX_ENOMEM (-3)  /* scratch allocation failed                      */
#define RLX_EUNSUP (-4)  /* shape/arch outside the supported envelope      */

#define RLX_ACT_TANH 0
#define RLX_ACT_ELU 1
#define RLX_ACT_RELU 2
#define RLX_ACT_NONE 3 /* identity (internal GEMM stages; not a valid rlx_mlp_desc.act) */
#define RLX_ACT_SILU 4 /* x * sigmoid(x): only behind a LayerNorm (rlx_lnmlp_desc, FastSAC); not a valid rlx_mlp_desc.act */

#define RLX_THREEFRY_LEGACY 0        /* jax_threefry_partitionable=False */
#define RLX_THREEFRY_PARTITIONABLE 1 /* default since JAX 0.5.0          */

typedef struct rlx_ctx rlx_ctx; /* opaque: device id + scratch arenas */

/* One MLP actor or critic.
 * arch A: rl_x/algorithms/ppo/flax/policy.py:31-40, critic.py:22-30
 *         n_hidden=2, hidden={H,H}, act=TANH, ln_first=0
 * arch B: rl_x/algorithms/ppo/flax_full_jit/policy.py:30-42, critic.py:21-32
 *         n_hidden=3, hidden={512,256,128}, act=ELU, ln_first=1                */
typedef struct rlx_mlp_desc {
  int32_t in_dim;
  int32_t n_hidden; /* 1..3 */
  int32_t hidden[4];
  int32_t out_dim;
  int32_t act;        /* RLX_ACT_*                                   */
  int32_t ln_first;   /* 1: LayerNorm(eps 1e-6) after the first Dense */
  int32_t has_logstd; /* policy: trailing logstd[out_dim]             */
} rlx_mlp_desc;

/* PPO hyper-parameters of one update (rl_x/algorithms/ppo/flax/default_config.py:9-26). */
typedef struct rlx_ppo_hparams {
  float clip_range;
  float entropy_coef;
  float critic_coef;
  float max_grad_norm; /* <= 0: no clipping */
  float adam_b1, adam_b2, adam_eps;
  int32_t discrete_actions; /* 0: diagonal-Gaussian policy (out_dim = action dim, has_logstd = 1).  1: Categorical policy
                             * (DiscreteFlatValuesPolicy, rl_x/algorithms/ppo/pytorch/policy.py:96-135 -- the only discrete
                             * PPO head of the reference): out_dim = number of actions (<= 8), has_logstd = 0, the `actions`
                             * arrays hold ONE float per sample = the action index. */
  const float* critic_states; /* NULL: policy and critic read the same observation rows (`states`, width pdesc->in_dim ==
                             * cdesc->in_dim).  Otherwise DEVICE [T, N, cdesc->in_dim]: the critic's OWN observation columns of
                             * the same rollout rows -- an env with critic_observation_indices != policy_observation_indices
                             * (`x[..., self.critic_observation_indices]`, rl_x/algorithms/ppo/flax/critic.py:12,24; policy.py:13,33);
                             * `states` then holds the policy's columns [T, N, pdesc->in_dim].  The column selection itself is
                             * rlx_select_columns_f32, applied once per acting step when the row is stored. */
} rlx_ppo_hparams;

/* ---- library ----------------------------------------------------------------- */
int rlx_version(void);
const char* rlx_last_error(void);
int rlx_ctx_create(int device, rlx_ctx** out);
int rlx_ctx_destroy(rlx_ctx* ctx);
/* number of fp32 parameters of `desc` (== oracle MLPSpec.n_params) */
int64_t rlx_mlp_param_count(const rlx_mlp_desc* desc);
/* out[m, j] = x[m, cols[j]] for j < n_cols: `x[..., indices]` of the networks that read a subset of the env's observation
 * (policy_observation_indices / critic_observation_indices: rl_x/algorithms/ppo/flax/policy.py:13,33, critic.py:12,24,
 * sac/flax/policy.py:14,31, critic.py:11,23, ppo_lstm/flax_full_jit/policy.py:15,74).  x: DEVICE [M, ldx]; cols: DEVICE
 * int32[n_cols], every entry in [0, ldx); out: DEVICE [M, ldo], ldo >= n_cols (columns >= n_cols are left untouched).      */
int rlx_select_columns_f32(rlx_ctx*, const float* x, int ldx, const int32_t* cols, int n_cols, float* out, int ldo, int64_t M,
                           void* stream);


/* ---- live kernel timing for bench.py's roofline leg ------------------------------------
 * Between rlx_prof_begin and rlx_prof_end every launch of the MFMA kernels is bracketed by HIP
 * events ON THE STREAM IT IS LAUNCHED ON.  rlx_prof_end synchronises the device and returns, per
 * kernel KIND k < rlx_prof_kernel_count() (names: rlx_prof_kernel_name(k) = "k_gemm_fwd",
 * "k_gemm_dx", "k_gemm_dw", "k_dx_l1bwd", "k_l1fwd_mfma", "k_head_loss", "k_reduce_segments"; a kind covers both engines, e.g. "k_gemm_fwd" =
 * k_gemm_fwd<> and k_gemm_bx<0,...>): total milliseconds, total algorithmic FLOPs
 * (2*M*N*K per launch), total algorithmic HBM bytes (every operand once) and launch count.      */
/* rlx_dbg_set_option("prof_sample", n): only every n-th launch of each (kernel, engine, shape) row carries events (default
 * 1 = all); averages and rates are then over the sampled launches, rlx_prof_union_ms is meaningful for n == 1 only.    */
int rlx_prof_kernel_count(void);
const char* rlx_prof_kernel_name(int k);
int rlx_prof_begin(rlx_ctx* ctx);
int rlx_prof_end(rlx_ctx* ctx, double* ms_out, double* flops_out, double* bytes_out, int64_t* count_out);
/* after rlx_prof_end: the same records per (kernel kind, engine, problem shape).  `launches` counts EVERY launch of the row
 * between begin and end, `timed` those that carried events (every prof_sample-th launch of THAT ROW: the sampling counter is
 * per row, so it cannot alias with a periodic launch pattern); ms / flops / bytes are sums over the timed launches.
 * engine: 0 = exact-fp32 MFMA kernels (k_gemm_fwd / k_gemm_dx / k_gemm_dw, k_dx_l1bwd<.., false>), 1 = split-fp32 operands on
 * the fp16 pipe (k_gemm_bx<0> / k_gemm_bx<1> / k_gemm_dw_bx, k_dx_l1bwd<.., true>).  Returns the row count in *n_out
 * (rows beyond `capacity` are not written).                                                                            */
typedef struct rlx_prof_row {
  int32_t kernel;   /* index for rlx_prof_kernel_name */
  int32_t engine;
  int32_t N, K;     /* GEMM shape of the launch: C[M, N] over a contraction of K (weight gradient: C[N.., ..] see DESIGN.md) */
  int64_t M;
  int64_t launches, timed;
  double ms, flops, bytes;
} rlx_prof_row;
int rlx_prof_rows(rlx_ctx* ctx, rlx_prof_row* rows, int capacity, int* n_out);
/* after rlx_prof_end: milliseconds during which AT LEAST ONE instrumented kernel was running (union of the launch
 * intervals over all streams) -- with policy and critic on two streams the per-launch durations overlap. */
int rlx_prof_union_ms(rlx_ctx* ctx, double* out);

/* test hook: named library options (an unknown name is RLX_EINVAL).
 * "disable_l1fused" = 1 routes the first-layer backward through the unfused kernels (k_gemm_dx + k_l1<bwd> +
 *   k_gemm_dw_skinny) so both paths stay tested.
 * "l1fwd_mfma" = 0: first-layer forward of the 512-wide LayerNorm / ELU shape on the VALU kernel instead of k_l1fwd_mfma.
 * "two_streams" = 0 makes the update calls run their networks back to back on the caller's stream instead of concurrently
 *   (PPO: critic on a library-owned side stream; SAC: the policy-loss chain).
 * "pipeline_updates" = 0 restores the join between consecutive minibatch updates of rlx_ppo_update_f32 (policy and critic
 *   chains otherwise run through the whole call without meeting; the gathered rows are double buffered).
 * "fused_recurrent_act" = 0 makes rlx_ppo_lstm_act_f32 use separate launches for torso / head / sampling / critic
 *   instead of the fused decoder kernel.
 * "gemm_bx" (default 1; environment RLX_GEMM_BX=0 sets the default of new contexts to 0): the hidden-layer GEMMs of passes
 *   with >= 4096 rows run on the half-precision matrix pipe with split fp32 operands (rl-x_amd/csrc/gemm_bx.h; fp64-referenced
 *   error budget in tests/test_gpu_gemm.py); 0 = the exact-fp32 MFMA engine everywhere.
 * "bx_debug" (also read from the environment variable RLX_BX_DEBUG when the context is created): bit 16 / 32 / 64 / 128 keeps the forward / input-gradient / weight-gradient / fused first-layer-backward
 *   kernels on the exact engine, bit 256 the recurrent product of k_lstm_seq_fwd.  "bx_force_mi" = 1 / 2 forces the 64- /
 *   128-row block tile of the split-operand kernels.  rlx_dbg_gemm_f32 modes 3 / 4 / 5 run the split forms of modes 0 / 1 / 2.
 * "adam_emit" (default 1): in rlx_ppo_update_f32 / rlx_ppo_update_dist_f32 the clip + Adam kernel rewrites the weight images
 *   from the parameters it has just written (bit-identical to laying them out again); 0 = one image launch per update and net.
 * "sac_twin" (default 1): both critics of a pair in ONE launch per layer (grid.y = 2; forward passes bit-identical to two
 *   sequential passes, weight gradients summed over half as many M-slabs).
 * "bx_gscale_log2" = e: the power-of-two scale 2^e the split-operand kernels apply to their GRADIENT operand when a test
 *   calls them outside an update (rlx_dbg_gemm_f32 modes 4 / 5); inside the update calls the library sets
 *   8 * 2^ceil(log2(global minibatch rows)) itself for the duration of every backward pass (gemm_bx.h: bx_grad_scale).
 * "prof_sample": see rlx_prof_begin.
 * Round 5 (PPO update; DESIGN.md section 4.2 has the measurements): "ppo_twin" (-1 default: policy || critic as twin launches,
 *   grid.y = 2 on one stream, for minibatches of 6144 to 16384 rows -- two chains on two streams, one all-reduce each, below (with
 *   grouped gathers) and above, and at every size when the context owns an RCCL communicator of more than one rank; 0 never; 1
 *   whenever the shapes allow), "ppo_tail" (-1 default: the last hidden layer forward + head + loss + both input gradients in one
 *   launch per network -- 32-row tiles up to 8192-row minibatches and above 16384, 64-row tiles in between; 0 off; 1 / 2 force a form), "l12_fused" (1: first + second layer forward in one launch),
 *   "dw_merge" (1: the weight gradients of two layers as one two-job launch), "dw_recompute" (0: the first-layer activations
 *   rebuilt inside the weight gradient instead of stored -- correct, fewer bytes, slower), "lf_idle_cus" (0).
 * Round 5 (SAC step): "fwd2h" (1: the whole forward of a 256-256 network incl. its head, and the dQ/da chain of the policy loss,
 *   as single launches per 32-row tile -- fwd2h.hip).  (What rlx_sac_update_f32 writes and keeps is NOT an option: see
 *   rlx_sac_hparams::keep_images and the NULL-able states / next_states arguments.)
 * Round 6: "gather_records" (1: a whole-update call first lays the rollout out as one aligned record per row, [obs | action |
 *   log_prob, return, advantage | pad] of 32 / 64 / 128 floats in a library arena (134 MB at T*N = 524288), and its minibatch
 *   gathers read two cache lines per sampled row instead of six; bit-identical results; 0: gather from the five arrays).
 *   "gather_group_rows" (524288: in the two-chain schedules of rlx_ppo_update_f32 / rlx_ppo_update_dist_f32 the rows of that many samples -- one epoch at
 *   configs[1] -- are gathered by ONE launch into one of two alternating buffers, and the chains meet once per group instead of once
 *   per update; 0: one gather per update; needs "gather_records").
 * (The measured-negative experiments of rounds 2-4 -- hipGraph replay, fused forward, 64-row / pipelined first-layer backward,
 *  split recurrent chains, plane-tensor GEMMs with direct-to-LDS staging, ... -- are documented in DESIGN.md Appendix A; their
 *  code lives in the git history only.)                                                                                    */
int rlx_dbg_set_option(rlx_ctx* ctx, const char* name, int value);
/* test hooks: "scratch_ptr:<bank>:<slot>" / "scratch_bytes:<bank>:<slot>" = device address / size of a library-owned scratch
 * arena (lets a test inspect intermediates); "allreduce_calls" = collectives this context has issued so far (through its RCCL
 * communicator or the all-reduce hook) -- bench.py's multi_gpu.collectives_per_step is the difference over the timed steps;
 * "bx_window_fallbacks" = rollouts (rlx_ppo_rollout_begin) that found a weight of the acting nets at or above 1023 -- outside
 * the fp16 window of the split-operand engine -- and therefore ran their T steps on the exact-fp32 engine; "gemm_bx" = the
 * current value of that option                                                                                          */
int rlx_dbg_get_counter(rlx_ctx* ctx, const char* name, int64_t* out);

/* test hook: the next rlx_sac_update_f32 calls take their N(0,1) draws from eps_next / eps_cur (DEVICE [B, A] each: the
 * next-state and current-state policy samples) instead of the per-sample threefry keys -- lets a test feed the update the
 * noise a reference run consumed.  NULL, NULL restores the PRNG.  The key schedule (key_io) is unaffected.            */
int rlx_dbg_set_sac_noise(rlx_ctx* ctx, const float* eps_next, const float* eps_cur);

/* tuning hook: instrumented kernels (fwd2h.hip: k_fwd2h) write clock64() stamps of their phases, taken by thread 0 of workgroup
 * (0, 0), to stamps[0 .. 16) (DEVICE, 16 x uint64).  NULL switches the stamps off.                                          */
int rlx_dbg_set_stamps(rlx_ctx* ctx, void* stamps);

/* debug / micro-benchmark hook: run ONE of the exact-fp32 MFMA GEMM kernels on caller buffers.
 *   mode 0: C[M,N]  = act(A[M,K] @ B[K,N] + aux[N])                 forward hidden layer
 *   mode 1: C[M,K] <- (A[M,N] @ B[K,N]^T) * act'(C[M,K]) in place   input gradient (act < 0: no act')
 *   mode 2: C[K,N]  = A[M,K]^T @ B[M,N]; aux[N] = column sums of B  weight gradient (split-M + reduce) */
int rlx_dbg_gemm_f32(rlx_ctx* ctx, int mode, const float* A, const float* B, float* C, float* aux, int64_t M, int N,
                     int K, int act, void* stream);

/* debug / micro-benchmark hook: the fused first layer (Dense + LayerNorm + activation) alone.
 * bwd = 0: H[M,Hd] = act(LN(X[M,O] @ W[O,Hd] + b));  bwd = 1: H holds dL/dH on entry and
 * dL/dZ1 on exit, ln_partials[grid][2*Hd] receives per-workgroup d(ln scale), d(ln bias).     */
int rlx_dbg_l1_f32(rlx_ctx* ctx, int bwd, const float* X, const float* W, const float* b, const float* g,
                   const float* be, float* H, float* ln_partials, int64_t M, int O, int Hd, int act, int ln, int grid,
                   void* stream);

/* ---- PRNG: jax.random restated (third-party jax<=0.7.2, not in the reference tree) --
 * call sites: rl_x/algorithms/ppo/flax/ppo.py:64-65,114-115,191-193                    */
/* host: `keys = jax.random.split(key, num)`; key_in/keys_out are HOST uint32 arrays     */
int rlx_threefry_split_host(const uint32_t key_in[2], uint32_t* keys_out /*[num,2]*/, int num, int scheme);
/* device: `_random_bits(key, 32, [n])` -> uint32[n]                                     */
int rlx_random_bits_u32(rlx_ctx*, const uint32_t key[2], uint32_t* out, int64_t n, int scheme, void* stream);
/* device: `jax.random.normal(key, [n])` -> float32[n] (Giles erfinv, like XLA f32)      */
int rlx_normal_f32(rlx_ctx*, const uint32_t key[2], float* out, int64_t n, int scheme, void* stream);
/* `key, sub = split(key); idx = permutation(sub, tile(arange(B),(E,1)), axis=1, independent=True)`
 * replaces rl_x/algorithms/ppo/flax/ppo.py:191-193.  key_io (HOST uint32[2]) is advanced in
 * place; out is DEVICE int32[E*B]; bit-exact w.r.t. jax (stable ascending sort per row,
 * ceil(3 ln(E*B)/ln(2^32-1)) rounds).                                                   */
int rlx_permutation_i32(rlx_ctx*, uint32_t key_io[2], int32_t* out, int E, int64_t B, int scheme, void* stream);

/* ---- synthetic random-observation env (the build's own) -------------
 * env object contract mirrored: rl_x/environments/custom_mujoco/ant/warp_torch/environment.py:142-186 */
int rlx_env_reset_f32(rlx_ctx*, uint32_t seed, int env_id_offset, int N, int obs_dim, int horizon,
                      float* obs /*[N,O]*/, int32_t* ep_step /*[N]*/, float* ep_ret /*[N]*/,
                      float* last_ret /*[N]*/, float* last_len /*[N]*/, void* stream);
int rlx_env_step_f32(rlx_ctx*, uint32_t seed, int env_id_offset, uint32_t t, int N, int obs_dim, int act_dim,
                     int horizon, float p_term, float reward_noise,
                     const float* action /*[N,A]*/, float* obs /*[N,O] in: current, out: post-reset next*/,
                     float* final_obs /*[N,O] pre-reset next obs (Batch.next_states row)*/,
                     float* reward /*[N]*/, float* terminated /*[N] 0/1*/, float* truncated /*[N] 0/1*/,
                     int32_t* ep_step, float* ep_ret, float* last_ret, float* last_len,
                     float* episode_stats /*dev [4] += {finished episodes, sum return, sum length, 0} or NULL*/,
                     void* stream);
/* the same step that also stores the PRE-step observation (the one the action was computed from) into prev_obs_out [N,O]
 * (NULL: none) -- the `states` row of the replay ring slot this transition fills (sac/flax/replay_buffer.py:23): one copy
 * launch less per vector step */
int rlx_env_step_copy_f32(rlx_ctx*, uint32_t seed, int env_id_offset, uint32_t t, int N, int obs_dim, int act_dim,
                          int horizon, float p_term, float reward_noise, const float* action, float* obs, float* final_obs,
                          float* reward, float* terminated, float* truncated, int32_t* ep_step, float* ep_ret,
                          float* last_ret, float* last_len, float* episode_stats, float* prev_obs_out, void* stream);

/* ---- acting: `get_action_and_value`, rl_x/algorithms/ppo/flax/ppo.py:110-119 ---------
 * (full-jit twin rl_x/algorithms/ppo/flax_full_jit/ppo.py:133-140).  key_io HOST uint32[2]
 * advanced like `key, subkey = split(key)`; noise = normal(subkey, [N_global,A]) rows
 * [env_id_offset, env_id_offset+N) so a sharded run draws the same noise as one device.
 * Also copies obs into states_row (Batch.states[t]) when states_row != NULL.             */
int rlx_actor_critic_fwd_sample_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams,
                                    const rlx_mlp_desc* cdesc, const float* cparams,
                                    const float* obs /*[N,O]*/, const float* critic_obs /*[N, cdesc->in_dim] or NULL: = obs*/,
                                    uint32_t key_io[2], int scheme,
                                    float* action /*[N,A]*/, float* processed /*[N,A] or NULL*/,
                                    float* value /*[N]*/, float* logp /*[N]*/, float* states_row /*[N,O] or NULL*/,
                                    int N, int clip_and_rescale, const float* act_low /*dev [A] or NULL*/,
                                    const float* act_high, int env_id_offset, int N_global, void* stream);
/* Categorical twin of rlx_actor_critic_fwd_sample_f32 for discrete action spaces (DiscreteFlatValuesPolicy,
 * rl_x/algorithms/ppo/pytorch/policy.py:96-135 -- the reference's only discrete PPO head; BASELINE.json configs[0],
 * CartPole): pdesc->out_dim = number of actions (2..8), has_logstd = 0.  action [N] = sampled index as float
 * (argmax(logits + Gumbel noise), the jax.random.categorical construction on the threefry stream; deterministic != 0:
 * argmax(logits), key untouched), logp [N] = log_softmax(logits)[action], value [N].                                  */
int rlx_actor_critic_fwd_sample_discrete_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams,
                                             const rlx_mlp_desc* cdesc, const float* cparams, const float* obs /*[N,O]*/,
                                             uint32_t key_io[2], int scheme, float* action /*[N]*/, float* value /*[N]*/,
                                             float* logp /*[N]*/, float* states_row /*[N,O] or NULL*/, int N,
                                             int env_id_offset, int N_global, int deterministic, void* stream);
/* ---- fused acting step: ONE launch = policy fwd + critic fwd + sample + log-prob
 * (+ the synthetic env transition when fuse_env != 0).  Same semantics and key schedule as
 * rlx_actor_critic_fwd_sample_f32 followed by rlx_env_step_f32, i.e. one iteration of the acting
 * loop rl_x/algorithms/ppo/flax/ppo.py:275-296 (full-jit: ppo/flax_full_jit/ppo.py:130-153).
 * Observations are double buffered by the caller: obs_in = Batch.states[t] (read only),
 * obs_out = Batch.states[t+1] (post-reset next observation, written when fuse_env).
 * rlx_ppo_rollout_step_supported() tells whether the network shapes fit the fused kernel
 * (in_dim <= 32, hidden[0] % 64 == 0 and <= 512, later hidden layers 128 or 256 wide).      */
int rlx_ppo_rollout_step_supported(const rlx_mlp_desc* pdesc, const rlx_mlp_desc* cdesc);
/* Optional bracket around the T acting steps of one rollout: rlx_ppo_rollout_begin lays out the split-fp16 weight images of the
 * hidden layers l >= 1 of both nets (one small launch; gemm_bx option), and the rlx_ppo_rollout_step_f32 calls that follow with
 * the SAME parameter pointers run those layers on the fp16 matrix pipe (no LDS weight stage, no barrier in the K loop).
 * CONTRACT: the parameters must not change between begin and the last step that should use the images; every
 * parameter-updating entry point of this library (rlx_ppo_update_f32, rlx_ppo_update_dist_f32, rlx_clip_adam_step_f32) and
 * rlx_ppo_rollout_end drop them.  A caller that writes the parameter buffers itself (checkpoint load) must call
 * rlx_ppo_rollout_end (or begin again).  Without a begin the steps use the exact-fp32 layers.
 * Engine window: begin first takes max |parameter| of both nets (two small launches + ONE blocking 8-byte read per rollout); a
 * weight at or above 1023 would overflow its fp16 image, so in that case no images are laid out, the T steps run on the exact-fp32
 * layers and the counter "bx_window_fallbacks" (rlx_dbg_get_counter) is incremented.                                          */
int rlx_ppo_rollout_begin(rlx_ctx* ctx, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                          const float* cparams, void* stream);
int rlx_ppo_rollout_end(rlx_ctx* ctx);
int rlx_ppo_rollout_step_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                             const float* cparams, const float* obs_in /*[N,O]*/, float* obs_out /*[N,O] or NULL*/,
                             uint32_t key_io[2], int scheme, float* action /*[N,A]*/, float* processed /*[N,A] or NULL*/,
                             float* value /*[N]*/, float* logp /*[N]*/, int N, int clip_and_rescale,
                             const float* act_low, const float* act_high, int noise_row_offset, int N_global,
                             int fuse_env, uint32_t env_seed, int env_id_offset, uint32_t env_t, int horizon,
                             float p_term, float reward_noise, float* final_obs /*[N,O]*/, float* reward /*[N]*/,
                             float* terminated /*[N]*/, int32_t* ep_step, float* ep_ret, float* last_ret,
                             float* last_len, float* episode_stats /*[4] or NULL*/, void* stream);

/* The T iterations of the acting loop (rl_x/algorithms/ppo/flax/ppo.py:275-296) queued by ONE call: step t reads states[t] and
 * writes actions[t], values[t], logps[t], final_obs[t], rewards[t], terminated[t] and the post-reset next observation into
 * states[t+1] (obs_last for t = T-1); the env clock of step t is env_t0 + t; key_io advances by T splits.  Same launches, keys
 * and results as T calls of rlx_ppo_rollout_step_f32 with fuse_env = 1 and processed = NULL (bit-identical:
 * tests/test_gpu_rollout.py) -- what changes is the host: it leaves the loop after ~0.5 ms instead of ~3 ms of per-step binding
 * overhead, so whatever the caller queues next (rlx_ppo_prefetch_permutation, GAE, the update) reaches the GPU while it still
 * runs the rollout.                                                                                                          */
int rlx_ppo_rollout_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams, const rlx_mlp_desc* cdesc,
                        const float* cparams, float* states /*[T,N,O]*/, float* obs_last /*[N,O]*/, uint32_t key_io[2],
                        int scheme, float* actions /*[T,N,A]*/, float* values /*[T,N]*/, float* logps /*[T,N]*/, int T, int N,
                        int clip_and_rescale, const float* act_low, const float* act_high, int noise_row_offset, int N_global,
                        uint32_t env_seed, int env_id_offset, uint32_t env_t0, int horizon, float p_term, float reward_noise,
                        float* final_obs /*[T,N,O]*/, float* rewards /*[T,N]*/, float* terminated /*[T,N]*/, int32_t* ep_step,
                        float* ep_ret, float* last_ret, float* last_len, float* episode_stats /*[4] or NULL*/, void* stream);

/* generic MLP forward: out[n,out_dim] = net(x[n,in_dim]);  critic on next_states
 * (rl_x/algorithms/ppo/flax/ppo.py:129) and deterministic actions (:235-238).            */
int rlx_mlp_fwd_f32(rlx_ctx*, const rlx_mlp_desc* desc, const float* params, const float* x, float* out,
                    int64_t n, void* stream);

/* ---- `next_values = critic(next_states)` of calculate_gae_advantages (rl_x/algorithms/ppo/flax/ppo.py:129) for a rollout
 * whose values[t] = critic(states[t]) were produced with the SAME critic parameters: next_values[t] = values[t+1] wherever
 * next_states[t] == states[t+1] bit for bit (all rows but the final-observation ones, ppo.py:277-286); the remaining rows
 * and the last step go through the critic.  The row selection, its length and the scatter stay on the device (no host
 * synchronisation); the result equals rlx_mlp_fwd_f32 on all T*N rows bit for bit.  All arrays time-major [T,N(,O)].    */
int rlx_ppo_next_values_f32(rlx_ctx*, const rlx_mlp_desc* cdesc, const float* cparams, const float* states,
                            const float* next_states, const float* values, float* next_values, int T, int N, void* stream);

/* ---- the iteration's logged scalars (rl_x/algorithms/ppo/flax/ppo.py:215-216, 226-230, 300-307) reduced ON THE DEVICE in two
 * library launches: out12[0..9] = column means of the [n_updates, 10] per-update metric rows of rlx_ppo_update_f32,
 * out12[10] = explained variance 1 - var(returns - values) / (var(returns) + 1e-8) over the n = T * N rollout entries
 * (population variances, fp64 sums in a fixed order), out12[11] = mean(exp(logstd[0..act_dim))) (logstd NULL: 0, the
 * Categorical policy).  All pointers DEVICE; the caller copies the 12 floats to the host once per iteration.              */
int rlx_ppo_reduce_metrics_f32(rlx_ctx*, const float* metrics, int n_updates, const float* returns, const float* values,
                               int64_t n, const float* logstd, int act_dim, float* out12, void* stream);

/* ---- GAE: `calculate_gae_advantages`, rl_x/algorithms/ppo/flax/ppo.py:122-135 --------
 * all arrays [T,N]; masks with terminations only (no reset at truncation).               */
int rlx_gae_f32(rlx_ctx*, const float* rewards, const float* values, const float* next_values,
                const float* terminations, float* advantages, float* returns, int T, int N, float gamma,
                float gae_lambda, void* stream);

/* ---- one minibatch: gather + advantage normalisation + loss + grads ------------------
 * replaces `minibatch_update` up to value_and_grad, rl_x/algorithms/ppo/flax/ppo.py:196-210
 * with loss_fn :142-177.  idx: DEVICE int32[mb_local] (flattened i = t*N+n).  Advantages are
 * normalised with mean/std over the gathered minibatch (population std, + 1e-8).
 * grads: DEVICE, same flat layout as params (policy then separate critic buffer).
 * metrics: DEVICE float[8] = {pg_loss, critic_loss, entropy_loss, approx_kl, clip_fraction,
 * adv_mean, adv_std, reserved} (means over the GLOBAL minibatch; the combined loss is
 * pg - entropy_coef*entropy + critic_coef*critic, linear in these means).
 * stats_io: NULL, or DEVICE double[4] for the multi-GPU two-phase protocol:
 *   phase 0 gathers the local rows and writes {sum_adv, sum_adv2, count, 0}, then returns;
 *   the host all-reduces stats_io (and sums mb_local into mb_global); phase 1 consumes the
 *   reduced sums, uses 1/mb_global as the loss denominator and produces LOCAL gradient /
 *   metric contributions that the host all-reduces (sum).  phase 2 = gather + consume
 *   externally supplied global sums in one call (batched statistics).
 *   phase 3 / 4 = phase 2 in halves: 3 gathers and runs the POLICY net only (cgrads may be NULL),
 *   4 runs the CRITIC net on the rows phase 3 gathered (pgrads may be NULL; `metrics` receives only
 *   the critic loss) -- lets the host all-reduce the policy gradients while the critic computes.
 *   phase 5 = gather + consume the supplied sums ONLY (no net); phase 6 = POLICY net on the gathered rows; with
 *   5 -> {6 on one stream, 4 on another} the host runs the two nets concurrently (the critic half uses its own
 *   activation / slab arenas).
 *   Single GPU: stats_io = NULL (phase ignored).                                         */
int rlx_ppo_minibatch_fwd_bwd_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams, float* pgrads,
                                  const rlx_mlp_desc* cdesc, const float* cparams, float* cgrads,
                                  float* metrics, const float* states, const float* actions,
                                  const float* log_probs, const float* returns, const float* advantages,
                                  const int32_t* idx, int mb_local, int mb_global, double* stats_io, int phase,
                                  const rlx_ppo_hparams* hp, void* stream);

/* ---- optimizer: optax.chain(clip_by_global_norm, adam), ppo/flax/ppo.py:84-100,212-213 */
int rlx_grad_global_norm_f32(rlx_ctx*, const float* grads, int64_t n, float* norm_out /*dev [1]*/, void* stream);
/* step is 1-based (optax count+1); lr evaluated by the host (linear_schedule, ppo.py:76-80);
 * grad_norm_out (dev [1]) receives the pre-clip global norm (metric, ppo.py:215-216).
 * NON-FINITE GRADIENTS (every Adam kernel of this library: this entry point, the PPO / PPO+LSTM update calls,
 * rlx_sac_update_f32, the FastSAC updates): when a network's global gradient norm is not finite its step is SKIPPED on the device
 * -- parameters, both moments, a fused Polyak target update and the weight-image rewrite are all left as they were -- and the
 * non-finite norm is still reported in the metrics.  The host-side step counter (opt_count_io / `step`) advances regardless, so
 * the bias correction of later steps is that of the advanced count.  This DEVIATES from the reference, deliberately: optax / JAX
 * would write the NaN into the parameters, torch.optim in FastSAC would do the same and still run its Polyak step.  The plugins
 * check the reported norms once per logged interval and stop (or, ppo.hip, redo the update on the exact-fp32 engine first).  */
int rlx_clip_adam_step_f32(rlx_ctx*, float* params, const float* grads, float* m, float* v, int64_t n_params,
                           int64_t step, float lr, float max_grad_norm, float b1, float b2, float eps,
                           float* grad_norm_out, void* stream);

/* ---- whole `update`: rl_x/algorithms/ppo/flax/ppo.py:138-232 -------------------------
 * runs nr_epochs*nr_minibatches minibatch updates (permutation + the two calls above) on
 * `stream`.  opt_count_io (HOST int64) = optimizer steps so far, advanced by E*M.
 * lr_schedule: HOST float[E*M] learning rate per update (host evaluates linear_schedule).
 * metrics_out: DEVICE float[E*M, 10] = the 8 minibatch metrics + policy_grad_norm +
 * critic_grad_norm per update (the reference means them over updates, ppo.py:226).        */
/* optional: generate -- on the library's side stream, ordered after the PREVIOUS rlx_ppo_update_f32 of this context
 * (its last read of the permutation buffer; before the first update: after the work already queued on `stream`), i.e.
 * UNDER the rollout whether the caller queues it before or after this call -- the permutation the NEXT
 * rlx_ppo_update_f32 call will need.  key_at_update (HOST) = the key that call will receive: the current key
 * advanced by the T acting splits, which are data independent.  A later update whose key_io / nr_epochs / T*N /
 * scheme match consumes it (its stream waits for the generation to finish); otherwise it is discarded.        */
int rlx_ppo_prefetch_permutation(rlx_ctx*, const uint32_t key_at_update[2], int nr_epochs, int64_t B, int scheme,
                                 void* stream);
/* ======================= data-parallel job: one process per GPU, RCCL over xGMI =======================
 * The reference has no multi-device path; contract = BASELINE.json north_star + SURVEY.md 8(b)/(e).
 * Envs -- and with them every [T, N, .] rollout array -- are sharded over the ranks (rank g owns the GLOBAL env ids
 * [env_id_offset, env_id_offset + N_local)); parameters, Adam moments and the PRNG key are replicated; the minibatch
 * permutation (ppo/flax/ppo.py:191-194) is computed identically on every rank over the GLOBAL index space
 * i = t * N_global + n; each rank runs the minibatch kernels on its local rows, normalising advantages with the
 * all-reduced statistics and scaling the loss by 1 / minibatch_size(global); the flat gradient vectors are summed over
 * the ranks with ncclAllReduce issued BY THE LIBRARY (no host callback), then clip + Adam run redundantly.
 *
 * RCCL is bound at run time (the process must hold one RCCL: the one PyTorch-ROCm mapped): rlx_dist_load_rccl(path)
 * with path = <torch>/lib/librccl.so, NULL = search (already-mapped symbols, $RLX_RCCL_LIBRARY, librccl.so).        */
/* what the RCCL entry points were bound to: the path of the copy this process had already mapped (PyTorch-ROCm's), "(global
 * symbols)", or the candidate a fresh copy was loaded from when none was mapped; "" before the first bind.  The library never
 * loads a second RCCL next to a mapped one.                                                                              */
const char* rlx_dist_rccl_path(void);
int rlx_dist_load_rccl(const char* path);
/* rank 0: ncclGetUniqueId -> id_out (HOST, 128 bytes); the host broadcasts it to the other ranks (any transport)      */
int rlx_dist_unique_id(void* id_out);
/* context of rank `rank` of `world`; a non-NULL id creates the communicator (ncclCommInitRank, collective
 * over all ranks) and the stream all collectives of this context are issued on.  world == 1 with id == NULL is
 * rlx_ctx_create; world == 1 WITH an id is a one-rank job whose collectives are still enqueued on RCCL (tests).      */
int rlx_ctx_create_dist(int device, int rank, int world, const void* nccl_unique_id /*128 B, or NULL iff world == 1*/,
                        rlx_ctx** out);
int rlx_ctx_rank(const rlx_ctx* ctx, int* rank_out, int* world_out);
/* ranks of the context's RCCL communicator as RCCL itself reports them (ncclCommCount); 0 when the context owns no
 * communicator (single rank, or collectives routed through the test hook) -- bench.py prints it next to n_gpus          */
int rlx_dist_comm_count(const rlx_ctx* ctx, int* count_out);
/* sum buf[n] (DEVICE fp32) over the ranks in place, ordered after the work queued on `stream` and before what is queued
 * on it next (SURVEY.md 8(b) export set; the update below calls the same routine internally)                            */
int rlx_allreduce_grads(rlx_ctx*, float* buf, int64_t n, void* stream);
/* rows a rank-local minibatch is padded to: the launch shapes of the data-parallel update are fixed (mean + 6.5 sigma of
 * the local row count, rounded up to 128; == minibatch_size when N_local == N_global); padding rows carry zero weight    */
int rlx_dist_row_capacity(int minibatch_size_global, int n_local, int n_global);
/* K4b: restrict the global permutation to this rank.  perm: DEVICE int32 [n_minibatches, minibatch_size_global] global
 * flattened indices; lidx: DEVICE int32 [n_minibatches, cap] receives, per minibatch and in permutation order, the LOCAL
 * flattened indices t * N_local + (n - env_id_offset) of the rows that live here; counts: DEVICE int32 [n_minibatches].   */
int rlx_dist_local_rows_i32(rlx_ctx*, const int32_t* perm, int n_minibatches, int minibatch_size_global, int n_local,
                            int n_global, int env_id_offset, int cap, int32_t* lidx, int32_t* counts, void* stream);
/* Capacity overflows since the last call (blocking on `stream`, where the read and the reset are ordered behind the updates
 * issued on it; reading resets both counters).  *rows_any_rank: rows dropped by ANY rank, taken from slot 3 of the per-minibatch
 * statistics records that rlx_ppo_update_dist_f32 all-reduces anyway -- identical on every rank, so a job that treats non-zero
 * as an error (the excess rows were dropped; probability < 1e-10 per minibatch) fails on ALL ranks in the same iteration and no
 * rank is left waiting in a collective for one that raised.  *minibatches_this_rank (may be NULL): minibatches THIS rank
 * truncated, including calls of rlx_dist_local_rows_i32 outside an update -- rank-local, a diagnostic only.               */
int rlx_dist_overflow_count(rlx_ctx*, int* rows_any_rank, int* minibatches_this_rank, void* stream);
/* optional, under the rollout: permutation + local-row restriction of the NEXT rlx_ppo_update_dist_f32 on the library's
 * side stream (same contract as rlx_ppo_prefetch_permutation)                                                         */
int rlx_ppo_dist_prefetch(rlx_ctx*, const uint32_t key_at_update[2], int nr_epochs, int T, int n_local, int n_global,
                          int env_id_offset, int minibatch_size_global, int scheme, void* stream);
/* the whole `update` (ppo/flax/ppo.py:138-232) on ONE RANK: rollout arrays are this rank's shard [T, N_local, .];
 * minibatch_size is GLOBAL (T * N_global must be a multiple of it); key_io / opt_count_io / lr_schedule / metrics_out as
 * rlx_ppo_update_f32.  Collectives per call: 1 (advantage sums, fp64 [E*M, 4]) + 2 * E*M (gradients) + 1 (metrics).
 * metrics_out holds the GLOBAL values on every rank.  Policy chain on `stream`, critic chain on the side stream, no
 * join between updates.  With one rank (and no hook) no collective is issued and the results equal rlx_ppo_update_f32's. */
int rlx_ppo_update_dist_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                            const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv, const float* states,
                            const float* actions, const float* log_probs, const float* returns, const float* advantages,
                            int T, int n_local, int n_global, int env_id_offset, int nr_epochs, int minibatch_size,
                            uint32_t key_io[2], int scheme, int64_t* opt_count_io, const float* lr_schedule,
                            const rlx_ppo_hparams* hp, float* metrics_out, void* stream);
/* test hooks: (1) stand-in for the collectives -- fn(user, buf, n, dtype (0 fp32, 1 fp64), on_side_stream) must sum buf
 * over the (emulated / gloo) ranks in place with work queued on the caller's stream (on_side_stream = 0) or on
 * rlx_ctx_side_stream (= 1); NULL restores RCCL.  (2) rank / world of a context WITHOUT communicator (emulated ranks).  */
typedef int (*rlx_allreduce_fn)(void* user, void* buf, int64_t n, int dtype, int on_side_stream);
int rlx_dbg_set_allreduce_hook(rlx_ctx*, rlx_allreduce_fn fn, void* user);
int rlx_dbg_set_rank(rlx_ctx*, int rank, int world);
/* the library-owned side stream (hipStream_t) of this context, created on first use */
void* rlx_ctx_side_stream(rlx_ctx*);
int rlx_ppo_update_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                       const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv,
                       const float* states, const float* actions, const float* log_probs,
                       const float* returns, const float* advantages, int T, int N,
                       int nr_epochs, int minibatch_size, uint32_t key_io[2], int scheme,
                       int64_t* opt_count_io, const float* lr_schedule, const rlx_ppo_hparams* hp,
                       float* metrics_out, void* stream);

/* ======================================= SAC ==========================================
 * Networks (flat layout as above): policy = MLP with out_dim = 2*act_dim, head columns
 * [0,A) = mean, [A,2A) = raw log_std (rl_x/algorithms/sac/flax/policy.py:31-41), no logstd param;
 * critic = TWO MLPs back to back (Q0 then Q1), each in_dim = obs+act, out_dim = 1
 * (rl_x/algorithms/sac/flax/critic.py:17-53); log_alpha = one float (entropy_coefficient.py:5-11). */
typedef struct rlx_sac_hparams {
  float gamma, tau, target_entropy;
  float log_std_min, log_std_max;
  float lr_policy, lr_critic, lr_alpha; /* host evaluates the schedule (sac.py:79-87) */
  float adam_b1, adam_b2, adam_eps;
  int32_t key_schedule; /* 0: host-loop flavour, keys = split(key, 2B+1), noise keys interleaved (sac/flax/sac.py:195-197);
                         * 1: fully jitted flavour, keys = split(key, 2B+2), keys[1] = replay-sampling key, noise keys in two
                         *    contiguous blocks (sac/flax_full_jit/sac.py:273-275)                                           */
  const float* critic_states;      /* NULL, or DEVICE [B, qdesc->in_dim - act_dim]: the critics' own observation columns of the   */
  const float* critic_next_states; /* sampled transitions (`x[..., critic_observation_indices]`, sac/flax/critic.py:11,23);
                                    * states / next_states then hold the policy's columns [B, pdesc->in_dim] (policy.py:14,31) */
  /* Data parallel (one process per GPU): this rank holds rows [batch_row_offset, batch_row_offset + B) of a
   * global batch of batch_global samples.  0 = the B rows are the whole batch.  With batch_global > B the losses are scaled
   * by 1 / batch_global, sample i draws its noise from key index batch_row_offset + i of split(key, 2 * batch_global + 1),
   * and ONE all-reduce (the context's RCCL communicator, rlx_dist_*) sums [policy grads | critic grads | loss sums] before
   * the three Adam steps, which every rank applies redundantly: parameters stay bit-identical across ranks.            */
  int64_t batch_global;
  int64_t batch_row_offset;
  /* Optional replay source (all NULL / 0: the batch arguments of rlx_sac_update_f32 already hold the transitions).  With
   * ring_states != NULL the update first gathers transition i = ring[ring_idx1[i], ring_idx2[i]] exactly like
   * rlx_sac_replay_sample_f32 -- INTO the batch arguments (states, next_states, actions, rewards, terminations are then
   * outputs as well) -- in the launch that lays out the critics' input rows: one launch and one call less per update.  */
  const float *ring_states, *ring_next_states, *ring_actions, *ring_rewards, *ring_terminations; /* [capacity, nr_envs, .] */
  const int32_t *ring_idx1, *ring_idx2;                                                          /* DEVICE int32 [B]       */
  int32_t ring_nr_envs;
  /* keep_images != 0 (batches >= 4096 rows): the five networks' split fp16 weight images stay in a library-owned arena after this
   * call -- the optimizer launch rewrites them from the parameters / Polyak targets it stores -- and the next rlx_sac_update_f32 /
   * rlx_sac_act_f32 call on the SAME parameter pointers and descriptors uses them instead of laying the images out again.  By
   * passing keep_images != 0 the caller states, for THIS call, that pparams / qparams / qtarget have not been written by anybody
   * but rlx_sac_update_f32 since its previous keep_images call on this context; after writing them (load, broadcast, a test poking
   * weights) call rlx_sac_invalidate_images first.  0: images are laid out per call and none are kept (any kept ones are dropped). */
  int32_t keep_images;
} rlx_sac_hparams;

/* drops the weight images a keep_images update left on the context: call it after writing SAC parameter / target vectors from
 * outside rlx_sac_update_f32 (the next update / acting call lays them out again from the parameters).  Always RLX_OK.   */
int rlx_sac_invalidate_images(rlx_ctx*);

/* `ReplayBuffer.sample` gather (rl_x/algorithms/sac/flax/replay_buffer.py:30-38) from the
 * device-resident ring [capacity, nr_envs, .]; idx1/idx2 are DEVICE int32[B] (drawn on the host with
 * numpy's Generator exactly like the reference: rng.integers(size), rng.integers(nr_envs)).        */
int rlx_sac_replay_sample_f32(rlx_ctx*, const float* ring_states, const float* ring_next_states,
                              const float* ring_actions, const float* ring_rewards, const float* ring_terminations,
                              int nr_envs, int obs_dim, int act_dim, const int32_t* idx1, const int32_t* idx2, int64_t B,
                              float* states, float* next_states, float* actions, float* rewards, float* terminations,
                              void* stream);
/* the fully jitted flavour's index draw (sac/flax_full_jit/sac.py:273-282) on the device: replay_key = split(update_key,
 * 2B+2)[1]; idx1 = randint(replay_key, (B,), 0, size); idx2 = randint(replay_key, (B,), 0, nr_envs) -- the SAME key for
 * both, as the reference has it.  update_key (HOST) = the key the following rlx_sac_update_f32 (key_schedule = 1) receives. */
int rlx_sac_replay_draw_i32(rlx_ctx*, const uint32_t update_key[2], int scheme, int64_t B, int size, int nr_envs,
                            int32_t* idx1 /*dev [B]*/, int32_t* idx2 /*dev [B]*/, void* stream);
/* `get_action` (sac.py:119-125): key, sub = split(key); action = tanh(mean + std * normal(sub, [N_global, A])[rows]);
 * deterministic != 0: tanh(mean), key untouched (sac.py:217-221).                                   */
int rlx_sac_act_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs /*[N,O]*/,
                    uint32_t key_io[2], int scheme, float* action /*[N,A]*/, int N, float log_std_min,
                    float log_std_max, int deterministic, int row_offset, int N_global, void* stream);
/* rlx_sac_act_f32 plus, from the same launch, the action the env receives -- `get_processed_action`
 * (sac/flax/policy.py:44-48): processed[n, j] = low[j] + (clip(action[n, j], -1, 1) + 1) * half_range[j] with
 * half_range = 0.5 * (high - low)  (low, half_range: DEVICE [A]; processed: DEVICE [N, A]).                      */
int rlx_sac_act_processed_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, const float* pparams, const float* obs /*[N,O]*/,
                              uint32_t key_io[2], int scheme, float* action /*[N,A]*/, int N, float log_std_min,
                              float log_std_max, int deterministic, int row_offset, int N_global, const float* low,
                              const float* half_range, float* processed, void* stream);
/* the whole jitted `update` (sac.py:128-215): per-sample noise keys split(key, 2B+1), loss_fn, three
 * plain Adam steps, Polyak.  opt_count_io (HOST) = optimizer steps so far, advanced by one.
 * metrics_out: DEVICE float[10] = {q_loss, policy_loss, entropy_loss, entropy, alpha, q_value,
 * policy_grad_norm, critic_grad_norm, entropy_grad_norm, 0}.
 * With the ring source (hp->ring_states != NULL) the five batch arguments are OUTPUTS (the gathered transitions); states and
 * next_states may then BOTH be NULL -- "I do not want the sampled observation rows back" (12 of the gather's 44 MB at configs[3])
 * -- provided obs_dim > 32 (narrow observations are read by the policy from these very arrays: RLX_EINVAL).  actions, rewards,
 * terminations and every other output are bit-identical with and without them (tests/test_gpu_sac.py).                */
int rlx_sac_update_f32(rlx_ctx*, const rlx_mlp_desc* pdesc, float* pparams, float* pm, float* pv,
                       const rlx_mlp_desc* qdesc, float* qparams /*[2*nq]*/, float* qm, float* qv, float* qtarget,
                       float* log_alpha /*dev [1]*/, float* am, float* av, const float* states, const float* next_states,
                       const float* actions, const float* rewards, const float* terminations, int64_t B,
                       uint32_t key_io[2], int scheme, int64_t* opt_count_io, const rlx_sac_hparams* hp,
                       float* metrics_out, void* stream);

/* ---- FastSAC's observation normaliser (rl_x/algorithms/fastsac/pytorch/observation_normalizer.py:11-53) ----------------
 * State on the DEVICE: running_mean, running_var, running_std_dev fp32 [O] (initially 0, 1, 1) and count int64[1] (0).
 * update: batch mean / population variance of obs [B, O] merged into the running statistics with the reference's formula
 * (its squared-difference term is taken against the already updated mean, :44-49), running_std_dev = sqrt(running_var),
 * count += B; fp64 column sums in a fixed order (bit-reproducible).  On a data-parallel context (communicator or all-reduce
 * hook) the sums and the row count are all-reduced first: every rank merges the GLOBAL batch, the statistics stay replicated.
 * apply:  out = (obs - running_mean) / (running_std_dev + epsilon)   (epsilon 1e-8 in the reference; out may alias obs).
 * The plugin calls them where FastSAC does (fastsac.py:253-254 acting / evaluation without update, :288-289 the sampled states
 * and next states with update), behind sac.hip's flag enable_observation_normalization.                                 */
int rlx_obs_norm_update_f32(rlx_ctx*, const float* obs, int64_t B, int O, float* running_mean, float* running_var,
                            float* running_std_dev, int64_t* count, void* stream);
int rlx_obs_norm_apply_f32(rlx_ctx*, const float* obs, int64_t B, int O, const float* running_mean,
                           const float* running_std_dev, float epsilon, float* out, void* stream);

/* ---- FastSAC's distributional critic step (rl_x/algorithms/fastsac/pytorch/fastsac.py:144-213) --------------------------
 * From the four networks' logits [B, nr_atoms] (online critics on (s, a), target critics on (s', a')) and the transition:
 * categorical projection of the entropy-adjusted n-step target onto the support linspace(v_min, v_max, nr_atoms) (the
 * reference's two index_add_ passes in their sequential order: deterministic), the clipped-double-Q choice of the projection
 * with the smaller expectation (or each critic's own), q_loss = q1_loss + q2_loss with q_k_loss = -mean_b sum_j target_kj
 * log_softmax(q_k)_j, and d q_loss / d logits of both online critics (what their backward passes start from).
 * rewards, dones, truncations, effective_n_steps, next_log_probs: DEVICE float[B]; log_alpha: DEVICE float[1] (alpha = exp);
 * out4: DEVICE float[4] = {q_loss, q_min, q_max, 0} (extrema of the first projection's expectation, fastsac.py:212-213).      */
int rlx_c51_critic_loss_f32(rlx_ctx*, const float* q1_logits, const float* q2_logits, const float* q1_next_logits,
                            const float* q2_next_logits, const float* rewards, const float* dones, const float* truncations,
                            const float* effective_n_steps, const float* next_log_probs, const float* log_alpha, int64_t B,
                            int nr_atoms, float gamma, float v_min, float v_max, int clipped_double_q, float* d_q1_logits,
                            float* d_q2_logits, float* out4, void* stream);

/* ---- FastSAC: networks and update steps (rl_x/algorithms/fastsac/pytorch) ------------------------------------------------
 * rlx_lnmlp_desc: every hidden layer is Dense -> LayerNorm (torch.nn.LayerNorm: population variance, eps 1e-5) -> SiLU, then a
 * Dense head (policy.py:46-57 with a [mean | log_std] head of width 2 * act_dim; q_network.py:27-38 with nr_atoms outputs).
 * FLAT LAYOUT: per hidden layer W[in, out] row-major, b[out], ln_scale[out], ln_bias[out]; then head W[in, out_dim], b[out_dim].
 * Hidden widths: multiples of 64, at most 768.                                                                                */
typedef struct rlx_lnmlp_desc {
  int32_t in_dim;
  int32_t n_hidden; /* 1..4 */
  int32_t hidden[4];
  int32_t out_dim;
} rlx_lnmlp_desc;
int64_t rlx_lnmlp_param_count(const rlx_lnmlp_desc*);
/* head output [M, out_dim] of x [M, in_dim] (row stride ldx >= in_dim)                                                      */
int rlx_lnmlp_fwd_f32(rlx_ctx*, const rlx_lnmlp_desc*, const float* params, const float* x, int ldx, float* out, int64_t M,
                      void* stream);

typedef struct rlx_fastsac_hparams { /* fastsac/pytorch/default_config.py:12-33 */
  float gamma, tau, v_min, v_max, log_std_min, log_std_max, target_entropy;
  float lr_policy, lr_critic, lr_alpha, weight_decay, adam_b1, adam_b2, adam_eps; /* torch.optim.AdamW (fastsac.py:88-91) */
  float max_grad_norm;       /* -1 (the reference's default): none; else torch.nn.utils.clip_grad_norm_ on the critics' / the policy's
                              * gradients before their AdamW steps: g *= min(1, c / (norm + 1e-6)) (fastsac.py:129-130, :218-219);
                              * the reported gradient norm is the un-clipped one, as clip_grad_norm_ returns it              */
  int32_t nr_atoms;          /* 2..128 */
  int32_t clipped_double_q;  /* clipped_double_q_learning */
} rlx_fastsac_hparams;

/* ReplayBuffer.sample (fastsac/pytorch/replay_buffer.py:34-96) for GIVEN start rows idx_t and env columns idx_e (DEVICE int32
 * [B]; the reference draws them with torch.randint over [0, max_start) x [0, nr_envs), max_start = capacity once the ring is
 * full, else max(1, size - n_steps + 1); n_steps == 1: [0, size)).  Ring arrays: DEVICE [capacity, nr_envs, .], pos / size as the
 * reference's add() leaves them.  Outputs [B, .]: state and action of the start row, the n-step discounted reward up to the first
 * done, next state / done / truncation of the step the window ends on, and the number of steps that counted.                    */
int rlx_fastsac_replay_sample_f32(rlx_ctx*, const float* ring_states, const float* ring_next_states, const float* ring_actions,
                                  const float* ring_rewards, const float* ring_dones, const float* ring_truncations, int capacity,
                                  int nr_envs, int obs_dim, int act_dim, int n_steps, float gamma, int pos, int size,
                                  const int32_t* idx_t, const int32_t* idx_e, int64_t B, float* states, float* next_states,
                                  float* actions, float* rewards, float* dones, float* truncations, float* effective_n_steps,
                                  void* stream);
/* policy.get_action (policy.py:93-108): action [N, A] = tanh(mean + exp(log_std) eps) * action_scale, or tanh(mean) *
 * action_scale when deterministic; log_std = min + 0.5 (max - min) (tanh(raw) + 1) (policy.py:66-72).  obs: already normalised /
 * column-selected policy observations [N, pdesc->in_dim].  key, subkey = split(key); eps[n, j] = normal(bits(subkey, (n +
 * row_offset) * A + j of N_global * A)) -- the reference draws with torch's CUDA generator, which nothing else reproduces;
 * rlx_dbg_set_sac_noise(eps_next, .) injects a given noise.                                                                  */
int rlx_fastsac_act_f32(rlx_ctx*, const rlx_lnmlp_desc* pdesc, const float* pparams, const float* obs, const float* action_scale,
                        uint32_t key_io[2], int scheme, float* action, int N, int deterministic, int row_offset, int N_global,
                        const rlx_fastsac_hparams* hp, void* stream);
/* ONE critic_and_entropy_loss_fn call plus the Polyak update that follows it (fastsac.py:144-241, :323-327): next action and
 * log-prob from the policy (no gradient), target critics on (s', a'), categorical projection of the entropy-adjusted n-step
 * target (rlx_c51_critic_loss_f32), both online critics' backward, AdamW step of the 2 * nq critic parameters (one optimizer
 * state, as in the reference), entropy-coefficient loss and its AdamW step, target <- (1 - tau) target + tau params.
 * qparams / qm / qv / qtarget: the two critics back to back.  critic_states / critic_next_states: NULL, or the critics' own
 * observation columns [B, qdesc->in_dim - act_dim].  dones / truncations / effective_n_steps: as ReplayBuffer.sample returns
 * them (replay_buffer.py:34-96).  opt_count_io (HOST): optimizer steps of the critic / entropy optimizers so far, advanced by 1.
 * metrics_out: DEVICE float[8] = {q_loss, entropy_loss, q_min, q_max, entropy, critic_grad_norm, entropy_grad_norm (the
 * reference logs norm^2, fastsac.py:237), alpha before the step}.                                                           */
int rlx_fastsac_critic_update_f32(rlx_ctx*, const rlx_lnmlp_desc* pdesc, const float* pparams, const rlx_lnmlp_desc* qdesc,
                                  float* qparams, float* qm, float* qv, float* qtarget, float* log_alpha, float* am, float* av,
                                  const float* states, const float* next_states, const float* critic_states,
                                  const float* critic_next_states, const float* actions, const float* rewards, const float* dones,
                                  const float* truncations, const float* effective_n_steps, const float* action_scale, int64_t B,
                                  uint32_t key_io[2], int scheme, int64_t* opt_count_io, const rlx_fastsac_hparams* hp,
                                  float* metrics_out, void* stream);
/* ONE policy_loss_fn call (fastsac.py:106-141): loss = mean(alpha log_prob - q), q = (q1 + q2) / 2 or min(q1, q2) of the
 * critics' EXPECTED values sum_j softmax(logits)_j z_j; gradient through both critics to the action and on through the squashing
 * to the policy; AdamW step of the policy.  metrics_out: DEVICE float[3] = {policy_loss, alpha, policy_grad_norm}.           */
int rlx_fastsac_policy_update_f32(rlx_ctx*, const rlx_lnmlp_desc* pdesc, float* pparams, float* pm, float* pv,
                                  const rlx_lnmlp_desc* qdesc, const float* qparams, const float* log_alpha, const float* states,
                                  const float* critic_states, const float* action_scale, int64_t B, uint32_t key_io[2], int scheme,
                                  int64_t* opt_count_io, const rlx_fastsac_hparams* hp, float* metrics_out, void* stream);

/* =================================== PPO + LSTM =======================================
 * Recurrent policy (rl_x/algorithms/ppo_lstm/flax_full_jit/policy.py:32-142, "concat" and "film" decoders):
 *   lstm_obs_encode / obs_encode: Dense(E)+LN+ELU on obs; OptimizedLSTMCell(H); LN+ELU on h;
 *   torso Dense(D1)+LN+ELU, Dense(D2)+ELU, Dense(D3)+ELU; mean head; state-independent logstd.
 * FLAT LAYOUT: enc_l {W[O,E], b, ln_g, ln_b} | enc_o {same} (absent when share_encoder) |
 *   lstm Wi[E,4H] (gate blocks i,f,g,o; no bias), Wh[H,4H], bh[4H] | lstm_ln g[H], b[H] |
 *   [film W[H,2E], b[2E] iff combine = FILM] | torso1 W[E+H,D1] ([E,D1] with FiLM), b, ln_g, ln_b | torso2 W, b |
 *   torso3 W, b | head W[D3,A], b | logstd[A].
 * The critic is the feed-forward PPO critic (ppo_lstm/flax_full_jit/critic.py:18-33).           */
typedef struct rlx_lstm_policy_desc {
  int32_t obs_dim, act_dim;
  int32_t enc_dim;      /* obs_encoding_dim  (default_config.py)                   */
  int32_t lstm_hidden;  /* lstm_hidden_dim: 64 in this build                       */
  int32_t torso[3];     /* policy torso widths (512, 256, 128)                     */
  int32_t share_encoder;/* share_lstm_obs_encoder / share_gru_obs_encoder          */
  int32_t cell;         /* RLX_CELL_LSTM, or RLX_CELL_GRU: the PPO+GRU policy
                         * (rl_x/algorithms/ppo_gru/flax_full_jit/policy.py:33-141, nn.GRUCell, single carry h:
                         * c_io / c0 arguments are then ignored).  GRU flat layout of the cell block:
                         * Wi[E,3H] (gate blocks r,z,n), bi[3H], Wh_rz[H,2H], Wh_n[H,H], bhn[H]. */
  int32_t combine;      /* lstm_obs_combine_method (policy.py:95-100): RLX_COMBINE_CONCAT: torso input = [obs latent | cell
                         * latent] (E + H wide); RLX_COMBINE_FILM: gamma = Dense(E)(cell latent), beta = Dense(E)(cell latent),
                         * torso input = obs latent * gamma + beta (E wide).  FiLM flat layout: after lstm_ln comes
                         * film W[H, 2E] (columns [0,E) = gamma kernel, [E,2E) = beta kernel), b[2E]; torso1 W is [E, D1]. */
} rlx_lstm_policy_desc;
#define RLX_CELL_LSTM 0
#define RLX_CELL_GRU 1
#define RLX_COMBINE_CONCAT 0
#define RLX_COMBINE_FILM 1

int64_t rlx_lstm_policy_param_count(const rlx_lstm_policy_desc* desc);

/* acting half of `single_rollout` (ppo_lstm.py:137-146): key, sub = split(key); Policy.apply_one_step
 * (policy.py:121-131) on obs [N,O] with carry (c_io, h_io) [N,H] updated IN PLACE (NOT yet masked with
 * done: the caller multiplies by (1-done) after the env step, ppo_lstm.py:148-149 -> rlx_lstm_mask_carry_f32);
 * action = mean + std * normal(sub, [N_global, A])[rows], log_prob, processed action, critic value.
 * deterministic != 0 (test mode): action = mean, key untouched.                                        */
/* the recurrent counterpart of rlx_ppo_rollout_begin (same contract; rlx_ppo_rollout_end and the parameter-updating entry
 * points drop the images): the fused decoder of the acting steps that follow runs its hidden layers on the fp16 matrix pipe  */
int rlx_ppo_lstm_rollout_begin(rlx_ctx* ctx, const rlx_lstm_policy_desc* desc, const float* pparams, const rlx_mlp_desc* cdesc,
                               const float* cparams, void* stream);
int rlx_ppo_lstm_act_f32(rlx_ctx*, const rlx_lstm_policy_desc* desc, const float* pparams, const rlx_mlp_desc* cdesc,
                         const float* cparams, const float* obs, const float* critic_obs /* NULL: the critic reads obs; else DEVICE
                         [N, cdesc->in_dim], its own observation columns (critic_observation_indices, critic.py:12,23) */,
                         float* c_io, float* h_io, uint32_t key_io[2],
                         int scheme, float* action, float* processed, float* value, float* logp, int N,
                         int clip_and_rescale, const float* act_low, const float* act_high, int noise_row_offset,
                         int N_global, int deterministic, void* stream);
/* carry *= (1 - done[:, None])   (ppo_lstm.py:148-149); done = terminated | truncated as 0/1 floats */
int rlx_lstm_mask_carry_f32(rlx_ctx*, float* c_io, float* h_io, const float* terminated, const float* truncated,
                            float* done_out /*[N] or NULL*/, int N, int H, void* stream);
/* one sequence minibatch of `minibatch_update` (ppo_lstm.py:231-259): gathers envs env_idx[ne] (all T steps,
 * rollout arrays [T,N,.]; c0/h0 [N,H] = rollout_init_policy_carry), normalises the advantages over the minibatch,
 * forward_sequence + loss_fn (ppo_lstm.py:181-216) + BPTT.  pgrads/cgrads/metrics as rlx_ppo_minibatch_fwd_bwd_f32. */
int rlx_ppo_lstm_minibatch_fwd_bwd_f32(rlx_ctx*, const rlx_lstm_policy_desc* desc, const float* pparams, float* pgrads,
                                       const rlx_mlp_desc* cdesc, const float* cparams, float* cgrads, float* metrics,
                                       const float* states, const float* actions, const float* log_probs,
                                       const float* returns, const float* advantages, const float* dones,
                                       const float* c0, const float* h0, const int32_t* env_idx, int nr_minibatch_envs,
                                       int T, int N, const rlx_ppo_hparams* hp, void* stream);
/* the whole optimisation phase (ppo_lstm.py:222-263): env-index permutation [E,N] -> E*M minibatches of
 * minibatch_size // T envs, clip + Adam per minibatch.  metrics_out: DEVICE float[E*M, 10] as rlx_ppo_update_f32.
 * Data parallel (context created with world > 1): N = THIS RANK's envs, minibatch_size = the GLOBAL minibatch.
 * Every rank permutes its local env indices with the same replicated key and contributes minibatch_size / (T * world) envs
 * to each minibatch (global minibatch = union over the ranks, each env once per epoch).  Collectives, all issued by the
 * library: the fp64 advantage sums of all minibatches once per call, each network's gradient once per minibatch on that
 * network's stream, the metrics once per call; losses are scaled by 1 / minibatch_size and every rank applies the same clip +
 * Adam step to the reduced gradients.                                                                                    */
int rlx_ppo_lstm_update_f32(rlx_ctx*, const rlx_lstm_policy_desc* desc, float* pparams, float* pm, float* pv,
                            const rlx_mlp_desc* cdesc, float* cparams, float* cm, float* cv, const float* states,
                            const float* actions, const float* log_probs, const float* returns, const float* advantages,
                            const float* dones, const float* c0, const float* h0, int T, int N, int nr_epochs,
                            int minibatch_size, uint32_t key_io[2], int scheme, int64_t* opt_count_io,
                            const float* lr_schedule, const rlx_ppo_hparams* hp, float* metrics_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLX_HIP_H */
