"""ppo.hip -- PPO whose whole training iteration runs as hand-written gfx950 kernels.

Host loop = the reference's (rl_x/algorithms/ppo/flax/ppo.py:109-390; device-resident
structure of rl_x/algorithms/ppo/flax_full_jit/ppo.py:130-265): T acting steps -> GAE ->
E*M minibatch updates -> metrics.  Every array lives in HBM as a torch tensor; every
computation is a librlxhip.so call (include/rlx_hip.h).  There is no CPU fallback.

PRNG key schedule (needed for bit-exact minibatch permutations), host-loop
variant: K = PRNGKey(seed); K, policy_key, critic_key = split(K, 3) (ppo/flax/ppo.py:64-65);
per acting step K, sub = split(K) (:114); per update K, sub = split(K) (:191).
Initial weights use numpy's QR-based orthogonal init seeded from policy_key / critic_key:
flax's orthogonal initialiser is only distribution-matched, never bit-matched.

Multi-GPU (one process per GPU, torch.distributed/RCCL): envs are sharded over ranks, params /
Adam moments / key are replicated, the permutation is computed identically on every rank over
the GLOBAL index space; each rank runs the minibatch kernels on its local rows with the global
advantage statistics and 1/mb_global.  Everything after the rollout is ONE library call
(rlx_ppo_update_dist_f32): the library restricts the permutation to this rank's rows, all-reduces the
advantage sums once, issues one ncclAllReduce per network and update on its own communicator (RCCL,
no host callback), applies clip+Adam redundantly and all-reduces the metric sums once.
"""
import json
import logging
import math
import os
import time

import numpy as np

from rlx_amd.algorithms.ppo.hip.general_properties import GeneralProperties
from rlx_amd.environments.action_space_type import ActionSpaceType
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.plugin import MetricSink, adopt_checkpoint_config

rlx_logger = logging.getLogger("rl_x")

METRIC_NAMES = ["loss/policy_gradient_loss", "loss/critic_loss", "loss/entropy_loss", "policy_ratio/approx_kl",
                "policy_ratio/clip_fraction", "advantages/minibatch_mean", "advantages/minibatch_std",
                "policy/std_dev", "gradients/policy_grad_norm", "gradients/critic_grad_norm"]


def _orthogonal(rng, shape, scale):
    n_rows, n_cols = shape
    big, small = max(n_rows, n_cols), min(n_rows, n_cols)
    q, r = np.linalg.qr(rng.standard_normal((big, small)))
    q = q * np.sign(np.diag(r))
    if n_rows < n_cols:
        q = q.T
    return scale * q


def _layout(in_dim, hidden, out_dim, ln_first, has_logstd):
    """Offsets of the flat parameter layout of include/rlx_hip.h."""
    off, layers, d = 0, [], in_dim
    for li, h in enumerate(hidden):
        L = {"in": d, "out": h, "W": off}
        off += d * h
        L["b"] = off
        off += h
        if ln_first and li == 0:
            L["g"] = off
            off += h
            L["be"] = off
            off += h
        layers.append(L)
        d = h
    head = {"in": d, "out": out_dim, "W": off}
    off += d * out_dim
    head["b"] = off
    off += out_dim
    logstd = None
    if has_logstd:
        logstd = off
        off += out_dim
    return layers, head, logstd, off


def init_flat_params(rng, in_dim, hidden, out_dim, ln_first, has_logstd, head_scale, std_dev):
    """orthogonal(sqrt 2) trunk, orthogonal(head_scale) head, zero biases, LN scale 1, logstd = log(std_dev)
    (rl_x/algorithms/ppo/flax_full_jit/policy.py:30-41, critic.py:24-31)."""
    layers, head, logstd, n = _layout(in_dim, hidden, out_dim, ln_first, has_logstd)
    p = np.zeros(n, dtype=np.float64)
    for L in layers:
        p[L["W"]:L["W"] + L["in"] * L["out"]] = _orthogonal(rng, (L["in"], L["out"]), np.sqrt(2)).ravel()
        if "g" in L:
            p[L["g"]:L["g"] + L["out"]] = 1.0
    p[head["W"]:head["W"] + head["in"] * head["out"]] = _orthogonal(rng, (head["in"], head["out"]), head_scale).ravel()
    if has_logstd:
        p[logstd:logstd + out_dim] = np.log(std_dev)
    return p.astype(np.float32)


class PPO:
    def __init__(self, config, train_env, eval_env, run_path, writer):
        import torch
        from rlx_amd.hip import ACT_ELU, ACT_TANH, Ctx, PpoHparams, mlp_desc
        from rlx_amd.hip import lib as hiplib
        self.torch = torch
        self.hiplib = hiplib
        self.config = config
        self.train_env = train_env
        self.eval_env = eval_env
        self.writer = writer

        self.save_model = config.runner.save_model
        self.save_path = os.path.join(run_path, "models")
        self.track_console = config.runner.track_console
        self.track_tb = config.runner.track_tb
        self.track_wandb = config.runner.track_wandb
        self.seed = config.environment.seed
        self.total_timesteps = config.algorithm.total_timesteps
        self.nr_envs = int(config.environment.nr_envs)                  # GLOBAL
        self.learning_rate = config.algorithm.learning_rate
        self.anneal_learning_rate = config.algorithm.anneal_learning_rate
        self.nr_steps = int(config.algorithm.nr_steps)
        self.nr_epochs = int(config.algorithm.nr_epochs)
        self.minibatch_size = int(config.algorithm.minibatch_size)       # GLOBAL
        self.gamma = config.algorithm.gamma
        self.gae_lambda = config.algorithm.gae_lambda
        self.clip_range = config.algorithm.clip_range
        self.entropy_coef = config.algorithm.entropy_coef
        self.critic_coef = config.algorithm.critic_coef
        self.max_grad_norm = config.algorithm.max_grad_norm
        self.std_dev = config.algorithm.std_dev
        self.action_clipping_and_rescaling = config.algorithm.action_clipping_and_rescaling
        self.evaluation_frequency = config.algorithm.evaluation_frequency
        self.evaluation_episodes = config.algorithm.evaluation_episodes
        self.scheme = 1 if config.algorithm.threefry_partitionable else 0
        self.use_fused_rollout = bool(config.algorithm.get("fused_rollout", True))
        self.rollout_one_call = bool(config.algorithm.get("rollout_one_call", True))   # rlx_ppo_rollout_f32 instead of T step calls
        # debugging aid: run the multi-GPU update protocol (sharded indices, batched statistics, all-reduce)
        # even with one rank, so the whole code path is exercised on a single GPU
        self.force_distributed_update = bool(config.algorithm.get("force_distributed_update", False))
        self.batch_size = self.nr_envs * self.nr_steps
        self.nr_updates = int(self.total_timesteps // self.batch_size)
        self.nr_minibatches = self.batch_size // self.minibatch_size

        if self.evaluation_frequency % (self.nr_steps * self.nr_envs) != 0 and self.evaluation_frequency != -1:
            raise ValueError("Evaluation frequency must be a multiple of the number of steps and environments.")
        if config.algorithm.device != "gpu":
            raise ValueError("ppo.hip runs on MI355X only: --algorithm.device must be 'gpu' (no CPU fallback)")
        if self.batch_size % self.minibatch_size != 0 or self.nr_minibatches < 1:
            raise ValueError("nr_envs * nr_steps must be a positive multiple of minibatch_size")
        self.host_env = train_env.general_properties.data_interface_type == DataInterfaceType.NUMPY
        if train_env.general_properties.data_interface_type not in (DataInterfaceType.TORCH, DataInterfaceType.NUMPY):
            raise ValueError("ppo.hip needs a TORCH or NUMPY data-interface environment")

        # distributed layout
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                self.dist = dist
        except Exception:
            pass
        self.nr_envs_local = getattr(train_env, "nr_envs", self.nr_envs // self.world)
        self.env_id_offset = getattr(train_env, "env_id_offset", self.rank * self.nr_envs_local)
        if self.nr_envs_local * self.world != self.nr_envs:
            raise ValueError("environment shard size * world size != environment.nr_envs")

        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = self._make_ctx(Ctx)
        self.sink = MetricSink(rlx_logger, writer, console=self.track_console, tensorboard=self.track_tb, wandb=self.track_wandb,
                               rank=self.rank)
        rlx_logger.info(f"Using device: {torch.cuda.get_device_name(self.device)} (rank {self.rank}/{self.world})")

        # PRNG: ppo/flax/ppo.py:64-65
        self.key = hiplib.prng_key(self.seed)
        ks = hiplib.threefry_split(self.key, 3, self.scheme)
        self.key, policy_key, critic_key = ks[0], ks[1], ks[2]

        self.os_shape = self.train_env.single_observation_space.shape
        self.as_shape = self.train_env.single_action_space.shape
        O, A = int(np.prod(self.os_shape)), int(np.prod(self.as_shape))
        # DISCRETE action spaces (BASELINE.json configs[0], CartPole): Categorical head as the reference's
        # DiscreteFlatValuesPolicy (ppo/pytorch/policy.py:96-135): logits of width env.get_single_action_logit_size(),
        # ONE stored value per action = its index, no log-std, no clipping / rescaling
        self.discrete = train_env.general_properties.action_space_type == ActionSpaceType.DISCRETE
        self._metrics12 = None       # device buffer of the iteration's 12 logged scalars (reduce_metrics)
        # Engine fallback (gemm_bx.h: the split-operand engine has an fp16 window -- |weight| < 1023, |hidden activation| < 4094,
        # per-sample gradient < 8190).  An update whose metrics come back non-finite on that engine is REDONE on the exact-fp32
        # engine from the state saved in front of it (the poisoned optimizer steps were skipped on the device anyway), the
        # context stays on the exact engine from then on and a warning is logged; only if the exact engine's result is
        # non-finite too does the run stop.  The acting nets are checked by the library before every rollout.
        self.engine_fallback = True
        self._snap = None
        self._bx_fallbacks_seen = 0
        if self.discrete:
            self.nr_actions = int(train_env.get_single_action_logit_size())
            if not 2 <= self.nr_actions <= 8 or A != 1:
                raise ValueError("ppo.hip: discrete action spaces need one action dimension with 2..8 choices")
            self.action_clipping_and_rescaling = False
            self.use_fused_rollout = False
        self.obs_dim, self.act_dim = O, A
        # Networks that read a SUBSET of the env's observation (`x[..., self.policy_observation_indices]`,
        # ppo/flax/policy.py:13,33, critic.py:12,24; full-jit: policy.py:15,32, critic.py:12,23): the selected columns are
        # stored per acting step (rlx_select_columns_f32) and the nets are built on the selected widths.
        pidx, cidx = self._observation_indices(train_env, O)
        self.obs_select = pidx is not None
        self.policy_obs_dim, self.critic_obs_dim = (len(pidx), len(cidx)) if self.obs_select else (O, O)
        if self.obs_select:
            if self.discrete:
                raise ValueError("ppo.hip: observation index sets are supported for continuous action spaces only")
            self.pidx = torch.from_numpy(pidx).to(self.device)
            self.cidx = torch.from_numpy(cidx).to(self.device)
            self.use_fused_rollout = False          # the fused acting kernel reads one shared observation row

        arch = config.algorithm.network_architecture
        if arch == "full_jit":
            hidden, act, ln = [512, 256, 128], ACT_ELU, True
        elif arch == "flax":
            h = int(config.algorithm.nr_hidden_units)
            hidden, act, ln = [h, h], ACT_TANH, False
        else:
            raise ValueError("algorithm.network_architecture must be 'full_jit' or 'flax'")
        head = self.nr_actions if self.discrete else A
        Op, Oc = self.policy_obs_dim, self.critic_obs_dim
        self.pdesc = mlp_desc(Op, hidden, head, act, ln, not self.discrete)
        self.cdesc = mlp_desc(Oc, hidden, 1, act, ln, False)
        prng = np.random.default_rng([int(policy_key[0]), int(policy_key[1])])
        crng = np.random.default_rng([int(critic_key[0]), int(critic_key[1])])
        pparams = init_flat_params(prng, Op, hidden, head, ln, not self.discrete, 0.01, self.std_dev)
        cparams = init_flat_params(crng, Oc, hidden, 1, ln, False, 1.0, self.std_dev)
        self.n_pparams, self.n_cparams = pparams.size, cparams.size
        self.logstd_offset = None if self.discrete else _layout(Op, hidden, A, ln, True)[2]
        dev = self.device
        self.pparams = torch.from_numpy(pparams).to(dev)
        self.cparams = torch.from_numpy(cparams).to(dev)
        self.pm, self.pv = torch.zeros_like(self.pparams), torch.zeros_like(self.pparams)
        self.cm, self.cv = torch.zeros_like(self.cparams), torch.zeros_like(self.cparams)
        self.opt_count = 0
        self.hp = PpoHparams(self.clip_range, self.entropy_coef, self.critic_coef, self.max_grad_norm, 0.9, 0.999, 1e-8)
        self.hp.discrete_actions = int(self.discrete)

        if self.discrete:          # a Discrete space has no bounds
            low = high = np.zeros(1, dtype=np.float32)
        else:
            low = np.asarray(self.train_env.single_action_space.low, dtype=np.float32).reshape(-1)
            high = np.asarray(self.train_env.single_action_space.high, dtype=np.float32).reshape(-1)
        self.act_low = torch.from_numpy(low).to(dev)
        self.act_high = torch.from_numpy(high).to(dev)

        if self.save_model:
            os.makedirs(self.save_path, exist_ok=True)
            self.best_mean_return = -np.inf
            self.best_model_file_name = "best.model"

    @staticmethod
    def _observation_indices(env, obs_dim):
        """(policy columns, critic columns) as int32 arrays when the env defines either index set, else (None, None).
        A missing set means all columns (`getattr(env, ..., jnp.arange(obs_dim))`, ppo/flax/policy.py:13)."""
        pidx = getattr(env, "policy_observation_indices", None)
        cidx = getattr(env, "critic_observation_indices", None)
        if pidx is None and cidx is None:
            return None, None
        out = []
        for name, idx in (("policy", pidx), ("critic", cidx)):
            idx = np.arange(obs_dim) if idx is None else np.asarray(idx).reshape(-1)
            if idx.size == 0 or idx.min() < 0 or idx.max() >= obs_dim:
                raise ValueError(f"{name}_observation_indices must be non-empty and within [0, {obs_dim})")
            out.append(np.ascontiguousarray(idx, dtype=np.int32))
        return out[0], out[1]

    # ------------------------------------------------------------------ schedule
    def lr_schedule(self):
        """linear_schedule (ppo/flax/ppo.py:76-80) for the next E*M optimizer steps."""
        n = self.nr_epochs * self.nr_minibatches
        if not self.anneal_learning_rate:
            return np.full(n, self.learning_rate, dtype=np.float32)
        counts = self.opt_count + np.arange(n)
        frac = 1.0 - (counts // (self.nr_minibatches * self.nr_epochs)) / max(self.nr_updates, 1)
        return (self.learning_rate * frac).astype(np.float32)

    # ------------------------------------------------------------------ buffers
    def _alloc_batch(self):
        t = self.torch
        T, N, O, A = self.nr_steps, self.nr_envs_local, self.obs_dim, self.act_dim
        f = dict(device=self.device, dtype=t.float32)
        B = type("Batch", (), {})()                          # rl_x/algorithms/ppo/flax/batch.py:1-11, time-major
        # with observation index sets: states = the POLICY's columns, cstates / next_states = the CRITIC's columns (only the
        # critic ever reads next_states, ppo/flax/ppo.py:122-123), next_full = one full-width row for the env to write into
        B.states = t.zeros(T, N, self.policy_obs_dim, **f)
        B.next_states = t.zeros(T, N, self.critic_obs_dim, **f)
        B.cstates = t.zeros(T, N, self.critic_obs_dim, **f) if self.obs_select else B.states
        B.next_full = t.zeros(N, O, **f) if self.obs_select else None
        B.actions = t.zeros(T, N, A, **f)
        B.rewards = t.zeros(T, N, **f)
        B.values = t.zeros(T, N, **f)
        B.terminations = t.zeros(T, N, **f)
        B.log_probs = t.zeros(T, N, **f)
        B.advantages = t.zeros(T, N, **f)
        B.returns = t.zeros(T, N, **f)
        B.next_values = t.zeros(T, N, **f)
        B.processed = t.zeros(N, A, **f)
        return B

    # ------------------------------------------------------------------ one iteration (device work only)
    def collect_rollout(self, batch, state):
        """T acting steps (ppo/flax/ppo.py:275-296).  Returns the next observation."""
        env, ctx = self.train_env, self.ctx
        if self.host_env:
            return self._collect_rollout_host(batch, state)
        fast = hasattr(env, "step_into")
        if (hasattr(env, "fused_args") and self.use_fused_rollout
                and ctx.rollout_step_supported(self.pdesc, self.cdesc)):
            return self._collect_rollout_fused(batch, state)
        for step in range(self.nr_steps):
            self._act(batch, state, step)
            if fast:
                env.step_into(batch.processed, batch.next_full if self.obs_select else batch.next_states[step],
                              batch.rewards[step], batch.terminations[step])
                if self.obs_select:
                    ctx.select_columns(batch.next_full, self.cidx, batch.next_states[step])
                state = env.obs
            else:
                next_state, reward, terminated, truncated, info = env.step(batch.processed)
                # TORCH interface: envs auto-reset; use their final observation when exposed
                fin = info.get("final_observation") if isinstance(info, dict) else None
                fin = fin if fin is not None else next_state
                if self.obs_select:
                    ctx.select_columns(fin.contiguous(), self.cidx, batch.next_states[step])
                else:
                    batch.next_states[step].copy_(fin)
                batch.rewards[step].copy_(reward)
                batch.terminations[step].copy_(terminated)
                state = next_state.contiguous()
        return state

    def _act(self, batch, state, step):
        """One acting call of the unfused loops: fills batch.actions/values/log_probs/states[step] and batch.processed."""
        ctx = self.ctx
        if self.discrete:
            self.key = ctx.actor_critic_fwd_sample_discrete(
                self.pdesc, self.pparams, self.cdesc, self.cparams, state, self.key, batch.actions[step].view(-1),
                batch.values[step], batch.log_probs[step], states_row=batch.states[step], scheme=self.scheme,
                env_id_offset=self.env_id_offset, n_global=self.nr_envs)
            batch.processed.copy_(batch.actions[step])        # the env receives the index unprocessed (ppo.py:236-240)
            return
        if self.obs_select:
            # x[..., indices] of both nets, stored where the update / GAE will read them; the nets run on the compact rows
            state = state.contiguous()
            ctx.select_columns(state, self.pidx, batch.states[step])
            ctx.select_columns(state, self.cidx, batch.cstates[step])
            self.key = ctx.actor_critic_fwd_sample(
                self.pdesc, self.pparams, self.cdesc, self.cparams, batch.states[step], self.key, batch.actions[step],
                batch.processed, batch.values[step], batch.log_probs[step], states_row=None,
                clip_and_rescale=self.action_clipping_and_rescaling, act_low=self.act_low, act_high=self.act_high,
                scheme=self.scheme, env_id_offset=self.env_id_offset, n_global=self.nr_envs, critic_obs=batch.cstates[step])
            return
        self.key = ctx.actor_critic_fwd_sample(
            self.pdesc, self.pparams, self.cdesc, self.cparams, state, self.key, batch.actions[step],
            batch.processed, batch.values[step], batch.log_probs[step], states_row=batch.states[step],
            clip_and_rescale=self.action_clipping_and_rescaling, act_low=self.act_low, act_high=self.act_high,
            scheme=self.scheme, env_id_offset=self.env_id_offset, n_global=self.nr_envs)

    def _collect_rollout_host(self, batch, state):
        """Host (NUMPY-interface) envs, the reference's acting loop (ppo/flax/ppo.py:275-296): per step ONE D2H copy of the
        processed actions and ONE packed H2D copy [next_state | final-obs-patched next_state | reward | terminated]
        through pinned staging buffers; everything else stays on the device."""
        t, env, ctx = self.torch, self.train_env, self.ctx
        N, O, A = self.nr_envs_local, self.obs_dim, self.act_dim
        if not hasattr(self, "_h_act"):
            self._h_act = t.empty(N, A, dtype=t.float32).pin_memory()
            self._h_pack = t.empty(N, 2 * O + 2, dtype=t.float32).pin_memory()
            self._d_pack = t.empty(N, 2 * O + 2, device=self.device)
            self._episode_stats = [0, 0.0, 0.0]
        pack = self._h_pack.numpy()
        for step in range(self.nr_steps):
            self._act(batch, state, step)
            self._h_act.copy_(batch.processed, non_blocking=True)
            t.cuda.current_stream().synchronize()          # the env needs the actions on the host
            next_state, reward, terminated, truncated, info = env.step(self._h_act.numpy())
            done = np.asarray(terminated) | np.asarray(truncated)
            pack[:, :O] = next_state
            pack[:, O:2 * O] = next_state                  # actual_next_state: final observation where the episode ended
            for i in np.flatnonzero(done):
                pack[i, O:2 * O] = np.asarray(env.get_final_observation_at_index(info, i))
                self._episode_stats[0] += 1
                self._episode_stats[1] += float(env.get_final_info_value_at_index(info, "episode_return", i))
                self._episode_stats[2] += float(env.get_final_info_value_at_index(info, "episode_length", i))
            pack[:, 2 * O] = reward
            pack[:, 2 * O + 1] = terminated
            self._d_pack.copy_(self._h_pack, non_blocking=True)
            t.cuda.current_stream().synchronize()          # the staging buffer is rewritten next step
            if self.obs_select:
                ctx.select_columns(self._d_pack[:, O:2 * O].contiguous(), self.cidx, batch.next_states[step])
            else:
                batch.next_states[step].copy_(self._d_pack[:, O:2 * O])
            batch.rewards[step].copy_(self._d_pack[:, 2 * O])
            batch.terminations[step].copy_(self._d_pack[:, 2 * O + 1])
            state = self._d_pack[:, :O].clone()
        return state

    def _collect_rollout_fused(self, batch, state):
        """One launch per acting step (rl-x_amd/csrc/rollout.hip).  Batch.states[t] doubles as the observation
        buffer: the env writes the next observation straight into states[t+1] (env.obs for the last step)."""
        env, ctx = self.train_env, self.ctx
        T = self.nr_steps
        if state.data_ptr() != batch.states[0].data_ptr():
            batch.states[0].copy_(state)
        # the parameters are constant for the T steps: lay out the acting nets' weight images once (fp16-pipe hidden layers)
        ctx.rollout_begin(self.pdesc, self.pparams, self.cdesc, self.cparams)
        if self.rollout_one_call:
            # the T launches queued by ONE library call: the host is out of the loop ~2.5 ms earlier than with the per-step
            # calls below, so the permutation prefetch and the update's launches reach the GPU while the rollout still runs
            self.key = ctx.rollout(
                self.pdesc, self.pparams, self.cdesc, self.cparams, batch.states, env.obs, self.key, batch.actions,
                batch.values, batch.log_probs, env.fused_args(batch.next_states, batch.rewards, batch.terminations),
                clip_and_rescale=self.action_clipping_and_rescaling, act_low=self.act_low, act_high=self.act_high,
                scheme=self.scheme, noise_row_offset=self.env_id_offset, n_global=self.nr_envs)
            env.fused_advance(T)
            ctx.rollout_end()
            return env.obs
        for step in range(T):
            obs_out = batch.states[step + 1] if step + 1 < T else env.obs
            self.key = ctx.rollout_step(
                self.pdesc, self.pparams, self.cdesc, self.cparams, batch.states[step], obs_out, self.key,
                batch.actions[step], None, batch.values[step], batch.log_probs[step],
                clip_and_rescale=self.action_clipping_and_rescaling, act_low=self.act_low, act_high=self.act_high,
                scheme=self.scheme, noise_row_offset=self.env_id_offset, n_global=self.nr_envs,
                env=env.fused_args(batch.next_states[step], batch.rewards[step], batch.terminations[step]))
            env.fused_advance()
        ctx.rollout_end()      # the weight images must not outlive the T steps (a load / an external optimizer may rewrite the parameters)
        return env.obs

    def compute_advantages(self, batch):
        """calculate_gae_advantages (ppo/flax/ppo.py:122-135): next_values = critic(next_states), then GAE -- two library
        calls.  The critic already evaluated states[t+1] during the rollout (same parameters), and next_states[t] equals
        states[t+1] bit for bit except where an episode ended (final-observation patch), so rlx_ppo_next_values_f32 sends
        only those rows and the last step through the critic again; the row selection never leaves the device."""
        self.ctx.ppo_next_values(self.cdesc, self.cparams, batch.cstates, batch.next_states, batch.values, batch.next_values)
        self.ctx.gae(batch.rewards, batch.values, batch.next_values, batch.terminations, batch.advantages,
                     batch.returns, self.gamma, self.gae_lambda)

    def update(self, batch, metrics_out):
        """update (ppo/flax/ppo.py:138-232)."""
        self.hp.critic_states = batch.cstates.data_ptr() if self.obs_select else None   # the critic's own observation columns
        if self.world == 1 and not self.force_distributed_update:
            self.key, self.opt_count = self.ctx.ppo_update(
                self.pdesc, self.pparams, self.pm, self.pv, self.cdesc, self.cparams, self.cm, self.cv,
                batch.states, batch.actions, batch.log_probs, batch.returns, batch.advantages, self.nr_epochs,
                self.minibatch_size, self.key, self.opt_count, self.lr_schedule(), self.hp, metrics_out, self.scheme)
        else:
            self._update_distributed(batch, metrics_out)

    def _distributed(self):
        return self.world > 1 or self.force_distributed_update

    def _make_ctx(self, Ctx):
        """One rlx_ctx per rank.  world > 1 over RCCL: rank 0 draws the communicator id inside the library and the host
        broadcasts its 128 bytes; the communicator lives in the context and every collective of the update is issued by
        the library.  Other torch.distributed backends (gloo: the 2-process tests on one GPU) go through the library's
        all-reduce hook instead."""
        t = self.torch
        if self.world == 1:
            return Ctx(self.device.index)
        dist = self.dist
        if dist.get_backend() == "nccl":
            ids = [self.hiplib.nccl_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            return Ctx(self.device.index, self.rank, self.world, ids[0])
        ctx = Ctx(self.device.index)
        ctx.set_rank(self.rank, self.world)
        side = ctx.side_stream()

        class _Buf:                                   # device memory handed over by the library -> torch view
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

        def hook(ptr, n, dtype, on_side):
            buf = t.as_tensor(_Buf(ptr, n, "<f8" if dtype else "<f4"), device=self.device)
            if on_side:
                with t.cuda.stream(side):
                    dist.all_reduce(buf)
            else:
                dist.all_reduce(buf)
        ctx.set_allreduce_hook(hook)
        return ctx

    def prefetch_permutation(self, rollout_queued=False):
        """The update's permutation depends only on the key after the T acting splits (all host-side, data independent),
        so it is generated on the library's side stream UNDER the rollout -- in the multi-GPU path together with the
        restriction to this rank's rows.  rollout_queued: the T acting steps have been issued already (self.key is the
        update's key): the ~250 sort launches then cost no GPU idle time in front of the first acting step."""
        k = self.key
        if not rollout_queued:
            for _ in range(self.nr_steps):
                k = self.hiplib.threefry_split(k, 2, self.scheme)[0]
        if not self._distributed():
            self.ctx.ppo_prefetch_permutation(k, self.nr_epochs, self.batch_size, self.scheme)
            return
        self.ctx.ppo_dist_prefetch(k, self.nr_epochs, self.nr_steps, self.nr_envs_local, self.nr_envs, self.env_id_offset,
                                   self.minibatch_size, self.scheme)

    def _update_distributed(self, batch, metrics_out):
        self.key, self.opt_count = self.ctx.ppo_update_dist(
            self.pdesc, self.pparams, self.pm, self.pv, self.cdesc, self.cparams, self.cm, self.cv, batch.states,
            batch.actions, batch.log_probs, batch.returns, batch.advantages, self.nr_envs, self.env_id_offset,
            self.nr_epochs, self.minibatch_size, self.key, self.opt_count, self.lr_schedule(), self.hp, metrics_out,
            self.scheme)

    def check_distributed_health(self):
        """Blocking.  A rank-local minibatch that outgrew its padded capacity dropped rows (probability < 1e-10 per
        minibatch).  The library sums the dropped rows of ALL ranks out of the statistics all-reduce of the same update, so
        every rank sees the same count here and the whole job raises in the same iteration -- no rank is left waiting in a
        collective for one that failed alone."""
        if self._distributed() and self.ctx.dist_overflow_count():
            raise RuntimeError("ppo.hip: a rank-local minibatch exceeded its row capacity on some rank (rows were dropped)")

    def reduce_metrics(self, batch, metrics_dev, allow_nonfinite=False):
        """The iteration's logged scalars (ppo/flax/ppo.py:215-216, 226-230, 300-307): mean of the E*M per-update metric rows,
        explained variance, policy std -- reduced ON THE DEVICE by two library launches (rlx_ppo_reduce_metrics_f32), then ONE
        device->host transfer per iteration (reference: per-step .cpu() calls, SURVEY.md call stack 2).  Returns the 12 host
        floats.  A non-finite loss or gradient norm stops the run here, naming the usual cause."""
        if self._metrics12 is None:
            self._metrics12 = self.torch.empty(12, device=self.device)
        logstd = None if self.discrete else self.pparams[self.logstd_offset:self.logstd_offset + self.act_dim]
        self.ctx.ppo_reduce_metrics(metrics_dev, batch.returns, batch.values, logstd, self._metrics12)
        host = self._metrics12.cpu().tolist()
        if not all(math.isfinite(v) for v in host[:10]):
            if allow_nonfinite:
                return None
            raise FloatingPointError(
                "ppo.hip: non-finite loss / gradient norm in this iteration " + str([round(v, 6) for v in host[:10]]) +
                ".  The optimizer steps of the affected updates were SKIPPED on the device (parameters and Adam moments hold "
                "their last finite values).  If the training itself is sane, an operand left the fp16 window of the split-operand "
                "GEMM engine (|hidden activation| >= 4094, |weight| >= 1023 or a per-sample gradient >= 8190; observations are "
                "scaled by their own maximum and cannot leave it; rl-x_amd/csrc/gemm_bx.h): rerun with RLX_GEMM_BX=0 "
                "(exact-fp32 MFMA engine).  [engine: " + ("split-operand" if self.ctx.get_counter("gemm_bx") else "exact fp32 "
                "already -- the training itself diverged") + "]")
        return host

    def train_iteration(self, batch, state, metrics_out, events=None):
        """One whole training iteration, exactly what train() runs per loop turn (and what bench.py times): T acting steps,
        GAE, E*M minibatch updates, the metric reduction and its one device->host copy.  events (optional): four
        torch.cuda.Event recorded around the three phases.  The host metric list lands in self.last_host_metrics."""
        if events:
            events[0].record()
        state = self.collect_rollout(batch, state)
        self.prefetch_permutation(rollout_queued=True)
        if events:
            events[1].record()
        self.compute_advantages(batch)
        if events:
            events[2].record()
        on_bx = self.engine_fallback and self.ctx.get_counter("gemm_bx") == 1
        if on_bx and self.ctx.get_counter("bx_window_fallbacks") != self._bx_fallbacks_seen:
            # the library found a weight of the acting nets outside the fp16 window and ran the rollout on the exact engine:
            # the update's weight images would overflow the same way
            self._bx_fallbacks_seen = self.ctx.get_counter("bx_window_fallbacks")
            self._to_exact_engine("a weight of the acting networks is at or above 1023")
            on_bx = False
        saved = self._save_update_state() if on_bx else None
        self.update(batch, metrics_out)
        if events:
            events[3].record()
        host = self.reduce_metrics(batch, metrics_out, allow_nonfinite=on_bx)
        if host is None:      # non-finite on the split-operand engine: the same update again, exact-fp32, from the saved state
            self._restore_update_state(saved)
            self._to_exact_engine("non-finite loss / gradient norm in this iteration's update")
            self.update(batch, metrics_out)
            host = self.reduce_metrics(batch, metrics_out)
        self.last_host_metrics = host
        return state

    def _to_exact_engine(self, why):
        self.ctx.set_option("gemm_bx", 0)
        rlx_logger.warning(
            "ppo.hip: %s -- an operand left the fp16 window of the split-operand GEMM engine (|weight| >= 1023, |hidden "
            "activation| >= 4094 or a per-sample gradient >= 8190; rl-x_amd/csrc/gemm_bx.h).  Training continues on the "
            "exact-fp32 MFMA engine (as with RLX_GEMM_BX=0): same results to 1e-5, about 1.8x the time per iteration (125 vs 71 ms at the bench shape).", why)

    def _save_update_state(self):
        """Everything rlx_ppo_update_f32 / _dist advances: both networks' parameters and Adam moments (six device copies of
        0.7 MB at the bench shape), the key and the optimizer step count."""
        src = (self.pparams, self.pm, self.pv, self.cparams, self.cm, self.cv)
        if self._snap is None:
            self._snap = [x.clone() for x in src]
        else:
            for d, x in zip(self._snap, src):
                d.copy_(x)
        return np.array(self.key, copy=True), self.opt_count

    def _restore_update_state(self, saved):
        for d, x in zip((self.pparams, self.pm, self.pv, self.cparams, self.cm, self.cv), self._snap):
            d.copy_(x)
        self.key, self.opt_count = saved

    # ------------------------------------------------------------------ training loop
    def train(self):
        t = self.torch
        self.set_train_mode()
        batch = self._alloc_batch()
        n_upd = self.nr_epochs * self.nr_minibatches
        metrics_dev = t.zeros(n_upd, 10, device=self.device)
        state, _ = self.train_env.reset()
        state = (t.from_numpy(np.ascontiguousarray(state, dtype=np.float32)).to(self.device) if self.host_env
                 else state.contiguous())
        global_step = 0
        nr_updates = 0
        nr_episodes = 0
        prev_end = None
        ev = [t.cuda.Event(enable_timing=True) for _ in range(4)]

        while global_step < self.total_timesteps:
            lr_now = float(self.lr_schedule()[0])
            state = self.train_iteration(batch, state, metrics_dev, ev)
            global_step += self.nr_steps * self.nr_envs
            nr_updates += n_upd
            host = self.last_host_metrics
            self.check_distributed_health()
            optimization_metrics = {METRIC_NAMES[i]: host[i] for i in (0, 1, 2, 3, 4, 8, 9)}
            optimization_metrics["lr/learning_rate"] = lr_now
            optimization_metrics["v_value/explained_variance"] = host[10]
            optimization_metrics["policy/std_dev"] = host[11]

            time_metrics = {
                "time/acting_time": ev[0].elapsed_time(ev[1]) / 1e3,
                "time/calc_adv_and_return_time": ev[1].elapsed_time(ev[2]) / 1e3,
                "time/optimizing_time": ev[2].elapsed_time(ev[3]) / 1e3,
            }
            # Evaluating (ppo/flax/ppo.py:323-344): deterministic actions on the eval env until evaluation_episodes finish
            evaluation_metrics = {}
            if self.evaluation_frequency != -1 and global_step % self.evaluation_frequency == 0:
                t_eval = time.time()
                self.set_eval_mode()
                rets, lens = self.evaluate(self.evaluation_episodes)
                evaluation_metrics = {"eval/episode_return": float(np.mean(rets)), "eval/episode_length": float(np.mean(lens))}
                self.set_train_mode()
                time_metrics["time/evaluating_time"] = time.time() - t_eval

            rollout_info_metrics = {}
            if self.host_env or hasattr(self.train_env, "pop_episode_stats"):
                if self.host_env:
                    n_done, sr, sl = self._episode_stats
                    mean_ret, mean_len = (sr / n_done, sl / n_done) if n_done else (float("nan"), float("nan"))
                    self._episode_stats = [0, 0.0, 0.0]
                else:
                    n_done, mean_ret, mean_len = self.train_env.pop_episode_stats()
                nr_episodes += n_done
                if n_done:
                    rollout_info_metrics = {"rollout/episode_return": mean_ret, "rollout/episode_length": mean_len}
                    if self.save_model and mean_ret > self.best_mean_return:
                        self.best_mean_return = mean_ret
                        self.save()
            now = time.time()
            if prev_end:
                time_metrics["time/sps"] = int((self.nr_steps * self.nr_envs) / (now - prev_end))
            prev_end = now

            steps_metrics = {"steps/nr_env_steps": global_step, "steps/nr_updates": nr_updates,
                             "steps/nr_episodes": nr_episodes}
            combined = {**rollout_info_metrics, **evaluation_metrics, **steps_metrics, **time_metrics, **optimization_metrics}
            self.sink.write(global_step, combined)
            self.last_metrics = combined

    # ------------------------------------------------------------------ checkpoint (native format)
    def save(self):
        if self.rank != 0:
            return
        path = os.path.join(self.save_path, self.best_model_file_name)
        state = {k: getattr(self, k).cpu().numpy() for k in ("pparams", "pm", "pv", "cparams", "cm", "cv")}
        np.savez(path + ".tmp.npz", opt_count=self.opt_count, policy_obs_dim=self.policy_obs_dim,
                 critic_obs_dim=self.critic_obs_dim, act_dim=self.act_dim,        # (rlx_amd/checkpoint.py reads them)
                 config_algorithm=json.dumps(self.config.algorithm.to_dict()), **state)
        os.replace(path + ".tmp.npz", path)

    def load(config, train_env, eval_env, run_path, writer, explicitly_set_algorithm_params):
        ckpt = np.load(config.runner.load_model, allow_pickle=False)
        adopt_checkpoint_config(config, json.loads(str(ckpt["config_algorithm"])), explicitly_set_algorithm_params)
        model = PPO(config, train_env, eval_env, run_path, writer)
        for k in ("pparams", "pm", "pv", "cparams", "cm", "cv"):
            getattr(model, k).copy_(model.torch.from_numpy(ckpt[k]).to(model.device))
        model.opt_count = int(ckpt["opt_count"])
        return model

    # ------------------------------------------------------------------ test mode (ppo/flax/ppo.py:469-485)
    def evaluate(self, episodes):
        """Deterministic (mean-action) episodes on the eval env, `get_deterministic_action` (ppo/flax/ppo.py:235-238).
        Returns (episode returns, episode lengths) of the first `episodes` finished episodes."""
        t = self.torch
        env = self.eval_env
        # copy_train_env_for_eval: the eval env IS the train env.  The reference's evaluation never perturbs the training
        # episodes, so the env state (observations, episode counters, RNG step, episode statistics) is saved here and
        # put back afterwards; an env that cannot do that must not be shared.
        shared = env is self.train_env
        if shared and not hasattr(env, "snapshot"):
            raise ValueError("ppo.hip: evaluation on the training env needs env.snapshot()/restore(); "
                             "set environment.copy_train_env_for_eval=False")
        snap = env.snapshot() if shared else None
        try:
            return self._evaluate(env, episodes)
        finally:
            if shared:
                env.restore(snap)

    def _evaluate(self, env, episodes):
        t = self.torch
        N, A = self.nr_envs_local, self.act_dim
        mean = t.empty(N, A, device=self.device)
        returns, lengths = [], []
        state, _ = env.reset()
        ep_ret = t.zeros(N, device=self.device)
        ep_len = t.zeros(N, device=self.device)
        to_dev = lambda a, dt=t.float32: t.from_numpy(np.ascontiguousarray(a)).to(self.device).to(dt)
        while len(returns) < episodes:
            if self.host_env:
                state = to_dev(np.asarray(state, dtype=np.float32))
            if self.discrete:                                                          # argmax(logits), ppo/pytorch/policy.py:133-135
                logits = t.empty(N, self.nr_actions, device=self.device)
                self.ctx.mlp_fwd(self.pdesc, self.pparams, state.contiguous(), logits)
                mean = logits.argmax(dim=1, keepdim=True).to(t.float32)
            else:
                obs = state.contiguous()
                if self.obs_select:
                    obs = self.ctx.select_columns(obs, self.pidx, t.empty(N, self.policy_obs_dim, device=self.device))
                self.ctx.mlp_fwd(self.pdesc, self.pparams, obs, mean)                   # deterministic action = mean
            action = mean
            if self.action_clipping_and_rescaling:
                action = self.act_low + 0.5 * (mean.clamp(-1, 1) + 1.0) * (self.act_high - self.act_low)
            if self.host_env:
                state, reward, terminated, truncated, info = env.step(action.cpu().numpy())
                reward, terminated, truncated = to_dev(reward), to_dev(terminated, t.bool), to_dev(truncated, t.bool)
            else:
                state, reward, terminated, truncated, info = env.step(action)
            ep_ret += reward
            ep_len += 1
            done = terminated | truncated
            if bool(done.any()):
                returns.extend(ep_ret[done].cpu().tolist())
                lengths.extend(ep_len[done].cpu().tolist())
                ep_ret = t.where(done, t.zeros_like(ep_ret), ep_ret)
                ep_len = t.where(done, t.zeros_like(ep_len), ep_len)
        return returns[:episodes], lengths[:episodes]

    def test(self, episodes):
        """ppo/flax/ppo.py:469-485."""
        self.set_eval_mode()
        returns, _ = self.evaluate(episodes)
        for i, r in enumerate(returns):
            rlx_logger.info(f"Episode {i + 1} - Return: {r}")
        return returns

    def set_train_mode(self):
        pass

    def set_eval_mode(self):
        pass

    def general_properties():
        return GeneralProperties
