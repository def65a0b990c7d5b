"""ppo_lstm.hip -- recurrent PPO whose training iteration runs as hand-written gfx950 kernels.

Mirrors rl_x/algorithms/ppo_lstm/flax_full_jit/ppo_lstm.py: single_rollout (:134-161, LSTM carry reset with
`done` after the env step), GAE (:169-185), sequence minibatches over an env-index permutation (:222-263),
evaluation with the mean action (:300-322).  All arrays live in HBM as torch tensors; every computation is a
librlxhip.so call (rlx_ppo_lstm_*_f32, include/rlx_hip.h).  There is no CPU fallback.

PRNG key schedule (:77-78, :117-118, :138, :226, :296): K = PRNGKey(seed); K, policy_key, critic_key, reset_key =
split(K, 4); train: K, seeds_key = split(K); seed key = split(seeds_key, 1)[0]; K', reset_key = split(seed key);
per evaluation block: K', sub = split(K') and the block runs on sub; per acting step and per optimisation phase
key, sub = split(key).
"""
import json
import logging
import os
import time

import numpy as np

from rlx_amd.algorithms.ppo.hip.ppo import PPO, METRIC_NAMES, _orthogonal, init_flat_params
from rlx_amd.algorithms.ppo_lstm.hip.general_properties import GeneralProperties
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.plugin import MetricSink, adopt_checkpoint_config

rlx_logger = logging.getLogger("rl_x")


def lstm_policy_layout(O, A, E, H, torso, share, cell="lstm", combine="concat"):
    """Offsets of the recurrent policy's flat parameter layout (include/rlx_hip.h, `rlx_lstm_policy_desc`)."""
    off, table = 0, {}

    def take(name, n):
        nonlocal off
        table[name] = (off, n)
        off += n
    D1, D2, D3 = torso
    for enc in (["enc_l"] if share else ["enc_l", "enc_o"]):
        take(enc + ".W", O * E); take(enc + ".b", E); take(enc + ".g", E); take(enc + ".be", E)
    if cell == "lstm":
        take("lstm.Wi", E * 4 * H); take("lstm.Wh", H * 4 * H); take("lstm.bh", 4 * H)
    else:
        take("gru.Wi", E * 3 * H); take("gru.bi", 3 * H); take("gru.Wh_rz", H * 2 * H); take("gru.Wh_n", H * H)
        take("gru.bhn", H)
    take("lstm_ln.g", H); take("lstm_ln.be", H)
    if combine == "film":
        take("film.W", H * 2 * E); take("film.b", 2 * E)
    take("t1.W", (E if combine == "film" else E + H) * D1); take("t1.b", D1); take("t1.g", D1); take("t1.be", D1)
    take("t2.W", D1 * D2); take("t2.b", D2)
    take("t3.W", D2 * D3); take("t3.b", D3)
    take("head.W", D3 * A); take("head.b", A)
    take("logstd", A)
    return table, off


def init_lstm_policy_params(rng, O, A, E, H, torso, share, std_dev, cell="lstm", combine="concat"):
    """Initialisers of policy.py:45-66: orthogonal(sqrt 2) Dense kernels, orthogonal(0.01) mean head, zero biases,
    LayerNorm scale 1; flax OptimizedLSTMCell defaults: lecun_normal input kernels, orthogonal recurrent kernels
    (one per gate), zero bias.  Distribution-matched to flax, never bit-matched."""
    table, n = lstm_policy_layout(O, A, E, H, torso, share, cell, combine)
    p = np.zeros(n, dtype=np.float64)

    def put(name, arr):
        o, m = table[name]
        p[o:o + m] = np.asarray(arr).ravel()
    D1, D2, D3 = torso
    for enc in (["enc_l"] if share else ["enc_l", "enc_o"]):
        put(enc + ".W", _orthogonal(rng, (O, E), np.sqrt(2)))
        put(enc + ".g", np.ones(E))
    # lecun_normal: truncated normal (+-2 sigma) with variance 1/fan_in
    lecun = lambda shape: np.clip(rng.standard_normal(shape), -2, 2) * (np.sqrt(1.0 / shape[0]) / 0.87962566103423978)
    if cell == "lstm":
        put("lstm.Wi", lecun((E, 4 * H)))
        put("lstm.Wh", np.concatenate([_orthogonal(rng, (H, H), 1.0) for _ in range(4)], axis=1))
    else:   # flax GRUCell defaults: lecun_normal input kernels, orthogonal recurrent kernels, zero biases
        put("gru.Wi", lecun((E, 3 * H)))
        put("gru.Wh_rz", np.concatenate([_orthogonal(rng, (H, H), 1.0) for _ in range(2)], axis=1))
        put("gru.Wh_n", _orthogonal(rng, (H, H), 1.0))
    put("lstm_ln.g", np.ones(H))
    if combine == "film":     # lstm_film_gamma / lstm_film_beta (policy.py:61-63): orthogonal(sqrt 2) kernels, zero biases
        put("film.W", np.concatenate([_orthogonal(rng, (H, E), np.sqrt(2)) for _ in range(2)], axis=1))
    put("t1.W", _orthogonal(rng, (E if combine == "film" else E + H, D1), np.sqrt(2)))
    put("t1.g", np.ones(D1))
    put("t2.W", _orthogonal(rng, (D1, D2), np.sqrt(2)))
    put("t3.W", _orthogonal(rng, (D2, D3), np.sqrt(2)))
    put("head.W", _orthogonal(rng, (D3, A), 0.01))
    put("logstd", np.full(A, np.log(std_dev)))
    return p.astype(np.float32), table


class PPO_LSTM(PPO):
    CELL = "lstm"            # flag-name prefix and recurrent cell; ppo_gru.hip derives from this class with "gru"

    def __init__(self, config, train_env, eval_env, run_path, writer):
        import torch
        from rlx_amd.hip import ACT_ELU, Ctx, PpoHparams, mlp_desc
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
        self.nr_parallel_seeds = config.algorithm.nr_parallel_seeds
        self.total_timesteps = config.algorithm.total_timesteps
        self.nr_envs = int(config.environment.nr_envs)
        self.learning_rate = config.algorithm.learning_rate
        self.anneal_learning_rate = config.algorithm.anneal_learning_rate
        self.nr_steps = int(config.algorithm.nr_steps)
        self.nr_epochs = int(config.algorithm.nr_epochs)
        self.minibatch_size = int(config.algorithm.minibatch_size)
        self.gamma = config.algorithm.gamma
        self.gae_lambda = config.algorithm.gae_lambda
        self.clip_range = config.algorithm.clip_range
        self.entropy_coef = config.algorithm.entropy_coef
        self.critic_coef = config.algorithm.critic_coef
        self.max_grad_norm = config.algorithm.max_grad_norm
        self.std_dev = config.algorithm.std_dev
        self.action_clipping_and_rescaling = config.algorithm.action_clipping_and_rescaling
        self.evaluation_and_save_frequency = config.algorithm.evaluation_and_save_frequency
        self.evaluation_active = config.algorithm.evaluation_active
        self.scheme = 1 if config.algorithm.threefry_partitionable else 0
        self.batch_size = self.nr_envs * self.nr_steps
        self.nr_updates = int(self.total_timesteps // self.batch_size)
        self.nr_minibatches = self.batch_size // self.minibatch_size
        self.nr_minibatch_envs = self.minibatch_size // self.nr_steps
        if config.algorithm.evaluation_and_save_frequency == -1:
            self.evaluation_and_save_frequency = self.batch_size * (self.total_timesteps // self.batch_size)
        self.evaluation_and_save_frequency = int(self.evaluation_and_save_frequency)
        self.horizon = getattr(self.train_env, "horizon", None)

        if self.evaluation_and_save_frequency % self.batch_size != 0:
            raise ValueError("Evaluation and save frequency must be a multiple of batch size")
        if self.nr_parallel_seeds > 1:
            raise ValueError("Parallel seeds are not supported yet. This is mainly limited by not being able to log mutliple wandb runs at the same time.")
        if config.algorithm.device != "gpu":
            raise ValueError("ppo_lstm.hip runs on MI355X only: --algorithm.device must be 'gpu' (no CPU fallback)")
        cell = self.CELL
        combine = config.algorithm[f"{cell}_obs_combine_method"]
        if combine not in ("concat", "film"):                        # policy.py:95-100
            raise ValueError(f"{cell}_obs_combine_method must be 'concat' or 'film'")
        self.combine = combine
        if self.minibatch_size % self.nr_steps != 0 or self.nr_minibatches < 1 or self.batch_size % self.minibatch_size != 0:
            raise ValueError("minibatch_size must be a multiple of nr_steps and divide nr_envs * nr_steps")
        if train_env.general_properties.data_interface_type != DataInterfaceType.TORCH:
            raise ValueError("ppo_lstm.hip needs a TORCH data-interface environment")

        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                self.dist = dist
        except Exception:
            pass
        # Data parallel (one process per GPU): envs sharded over the ranks, parameters / Adam moments / key
        # replicated.  Every rank permutes ITS env indices with the replicated key and contributes nr_minibatch_envs / world of
        # them to each minibatch (the global minibatch is the union; each env once per epoch); the library all-reduces the
        # advantage statistics of all minibatches once, each network's gradient once per minibatch, the metrics once
        # (rlx_ppo_lstm_update_f32 on a context with a communicator).
        self.nr_envs_local = int(getattr(train_env, "nr_envs", self.nr_envs // self.world))
        self.env_id_offset = int(getattr(train_env, "env_id_offset", self.rank * self.nr_envs_local))
        if self.nr_envs_local * self.world != self.nr_envs:
            raise ValueError("environment shard size * world size != environment.nr_envs")
        if self.nr_minibatch_envs % self.world != 0 or self.nr_envs_local % max(self.nr_minibatch_envs // self.world, 1) != 0:
            raise ValueError("minibatch_size / nr_steps must be divisible by the number of ranks and divide the rank's envs")
        self.use_fused_rollout = False
        self.force_distributed_update = False
        self.discrete, self._metrics12 = False, None      # (PPO.reduce_metrics)

        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = self._make_ctx(Ctx)                                  # RCCL communicator in the context (gloo tests: the hook)
        self.sink = MetricSink(rlx_logger, writer, console=self.track_console, tensorboard=self.track_tb, wandb=self.track_wandb,
                               rank=self.rank)
        rlx_logger.info(f"Using device: {torch.cuda.get_device_name(self.device)}")

        # ppo_lstm.py:77-78
        self.key = hiplib.prng_key(self.seed)
        ks = hiplib.threefry_split(self.key, 4, self.scheme)
        self.key, policy_key, critic_key = ks[0], ks[1], ks[2]

        self.os_shape = self.train_env.single_observation_space.shape
        self.as_shape = self.train_env.single_action_space.shape
        O, A = int(np.prod(self.os_shape)), int(np.prod(self.as_shape))
        self.obs_dim, self.act_dim = O, A
        # `obs[..., self.policy_observation_indices]` (ppo_lstm/flax_full_jit/policy.py:15,74, critic.py:12,23): as in ppo.hip the
        # selected columns are stored per acting step (rlx_select_columns_f32) and the nets are built on the selected widths --
        # batch.states holds the policy's columns, batch.cstates / next_states the critic's.
        pidx, cidx = self._observation_indices(train_env, O)
        self.obs_select = pidx is not None
        self.policy_obs_dim, self.critic_obs_dim = (len(pidx), len(cidx)) if self.obs_select else (O, O)
        if self.obs_select:
            self.pidx = torch.from_numpy(pidx).to(self.device)
            self.cidx = torch.from_numpy(cidx).to(self.device)
        Op, Oc = self.policy_obs_dim, self.critic_obs_dim
        E, H = int(config.algorithm.obs_encoding_dim), int(config.algorithm[f"{cell}_hidden_dim"])
        self.enc_dim, self.lstm_hidden = E, H
        torso = (512, 256, 128)
        share = bool(config.algorithm[f"share_{cell}_obs_encoder"])
        self.ldesc = hiplib.lstm_policy_desc(Op, A, E, H, torso, share, hiplib.CELL_GRU if cell == "gru" else hiplib.CELL_LSTM,
                                             hiplib.COMBINE_FILM if self.combine == "film" else hiplib.COMBINE_CONCAT)
        self.cdesc = mlp_desc(Oc, [512, 256, 128], 1, ACT_ELU, True, False)   # critic.py:18-33
        prng = np.random.default_rng([int(policy_key[0]), int(policy_key[1])])
        crng = np.random.default_rng([int(critic_key[0]), int(critic_key[1])])
        pparams, table = init_lstm_policy_params(prng, Op, A, E, H, torso, share, self.std_dev, cell, self.combine)
        cparams = init_flat_params(crng, Oc, [512, 256, 128], 1, True, False, 1.0, self.std_dev)
        if self.ctx.lstm_policy_param_count(self.ldesc) != pparams.size:
            raise RuntimeError("host / device parameter layouts disagree")
        self.n_pparams, self.n_cparams = pparams.size, cparams.size
        self.logstd_offset = table["logstd"][0]
        dev = self.device
        self.pparams = torch.from_numpy(pparams).to(dev)
        self.cparams = torch.from_numpy(cparams).to(dev)
        self.pm, self.pv = torch.zeros_like(self.pparams), torch.zeros_like(self.pparams)
        self.cm, self.cv = torch.zeros_like(self.cparams), torch.zeros_like(self.cparams)
        self.opt_count = 0
        self.hp = PpoHparams(self.clip_range, self.entropy_coef, self.critic_coef, self.max_grad_norm, 0.9, 0.999, 1e-8)
        # policy.initialize_carry (policy.py:69-71)
        self.carry_c = torch.zeros(self.nr_envs_local, H, device=dev)
        self.carry_h = torch.zeros(self.nr_envs_local, H, device=dev)

        low = np.asarray(self.train_env.single_action_space.low, dtype=np.float32).reshape(-1)
        high = np.asarray(self.train_env.single_action_space.high, dtype=np.float32).reshape(-1)
        self.act_low = torch.from_numpy(low).to(dev)
        self.act_high = torch.from_numpy(high).to(dev)

        if self.save_model:
            os.makedirs(self.save_path, exist_ok=True)
            self.best_mean_return = -np.inf
            self.best_model_file_name = "latest.model"

    # ------------------------------------------------------------------ buffers
    def _alloc_batch(self):
        t = self.torch
        B = super()._alloc_batch()
        T, N, H = self.nr_steps, self.nr_envs_local, self.lstm_hidden
        f = dict(device=self.device, dtype=t.float32)
        B.dones = t.zeros(T, N, **f)
        B.truncations = t.zeros(N, **f)
        B.c0 = t.zeros(N, H, **f)       # rollout_init_policy_carry (ppo_lstm.py:163)
        B.h0 = t.zeros(N, H, **f)
        return B

    # ------------------------------------------------------------------ one iteration
    def collect_rollout(self, batch, state):
        """single_rollout x nr_steps (ppo_lstm.py:134-164)."""
        env, ctx = self.train_env, self.ctx
        fast = hasattr(env, "step_into")
        batch.c0.copy_(self.carry_c)
        batch.h0.copy_(self.carry_h)
        ctx.ppo_lstm_rollout_begin(self.ldesc, self.pparams, self.cdesc, self.cparams)   # parameters are constant for the T steps
        sel = self.obs_select
        for step in range(self.nr_steps):
            if sel:     # x[..., indices] of both nets, stored where the update / GAE read them
                state = state.contiguous()
                ctx.select_columns(state, self.pidx, batch.states[step])
                ctx.select_columns(state, self.cidx, batch.cstates[step])
            else:
                batch.states[step].copy_(state)
            self.key = ctx.ppo_lstm_act(
                self.ldesc, self.pparams, self.cdesc, self.cparams, batch.states[step], self.carry_c, self.carry_h,
                self.key, batch.actions[step], batch.processed, batch.values[step], batch.log_probs[step],
                clip_and_rescale=self.action_clipping_and_rescaling, act_low=self.act_low, act_high=self.act_high,
                scheme=self.scheme, noise_row_offset=self.env_id_offset, n_global=self.nr_envs,
                critic_obs=batch.cstates[step] if sel else None)
            if fast:
                env.step_into(batch.processed, batch.next_full if sel else batch.next_states[step], batch.rewards[step],
                              batch.terminations[step], batch.truncations)
                if sel:
                    ctx.select_columns(batch.next_full, self.cidx, batch.next_states[step])
                state = env.obs
                ctx.lstm_mask_carry(self.carry_c, self.carry_h, batch.terminations[step], batch.truncations, batch.dones[step])
            else:
                next_state, reward, terminated, truncated, info = env.step(batch.processed)
                fin = info.get("final_observation") if isinstance(info, dict) else None
                fin = fin if fin is not None else next_state
                if sel:
                    ctx.select_columns(fin.contiguous(), self.cidx, batch.next_states[step])
                else:
                    batch.next_states[step].copy_(fin)
                batch.rewards[step].copy_(reward)
                batch.terminations[step].copy_(terminated)
                batch.truncations.copy_(truncated)
                ctx.lstm_mask_carry(self.carry_c, self.carry_h, batch.terminations[step], batch.truncations, batch.dones[step])
                state = next_state.contiguous()
        ctx.rollout_end()      # the acting nets' weight images must not outlive the T steps
        return state

    def update(self, batch, metrics_out):
        """ppo_lstm.py:222-263."""
        self.hp.critic_states = batch.cstates.data_ptr() if self.obs_select else None   # the critic's own observation columns
        self.key, self.opt_count = self.ctx.ppo_lstm_update(
            self.ldesc, self.pparams, self.pm, self.pv, self.cdesc, self.cparams, self.cm, self.cv, batch.states,
            batch.actions, batch.log_probs, batch.returns, batch.advantages, batch.dones, batch.c0, batch.h0,
            self.nr_epochs, self.minibatch_size, self.key, self.opt_count, self.lr_schedule(), self.hp, metrics_out,
            self.scheme)

    # ------------------------------------------------------------------ training loop
    def train(self):
        # ppo_lstm.py:357-358 and :117-118: the training key is derived from split(split(key)[1], nr_parallel_seeds)[0]
        ks = self.hiplib.threefry_split(self.key, 2, self.scheme)
        self.key = ks[0]
        run_key = self.hiplib.threefry_split(ks[1], 1, self.scheme)[0]
        ks = self.hiplib.threefry_split(run_key, 2, self.scheme)       # key, reset_key (reset keys drive the env only)
        outer_key = ks[0]
        saved_key, self.key = self.key, outer_key
        try:
            self._train_blocks()
        finally:
            self.key = saved_key

    def _train_blocks(self):
        per_block = self.evaluation_and_save_frequency // self.batch_size
        nr_blocks = int(self.total_timesteps // self.evaluation_and_save_frequency)
        if nr_blocks == 0:
            rlx_logger.warning("total_timesteps < evaluation_and_save_frequency: nothing to train (as in the reference); "
                               "set --algorithm.evaluation_and_save_frequency=-1 for a single block")
        total = self.total_timesteps
        start = time.time()
        self._loop_state = None
        for block in range(nr_blocks):
            # key, subkey = split(key); the block's learning iterations run on subkey and hand their key back (:296-298)
            ks = self.hiplib.threefry_split(self.key, 2, self.scheme)
            self.key = ks[1]
            self.total_timesteps = (block + 1) * per_block * self.batch_size
            self._run_learning_iterations(block * per_block * self.batch_size)
            if self.evaluation_active:
                self._evaluate((block + 1) * per_block * self.batch_size)
            if self.save_model:
                self.save()
        self.total_timesteps = total
        rlx_logger.info(f"Average time: {time.time() - start:.2f} s")

    def _run_learning_iterations(self, global_step):
        """PPO.train's loop body, resumable across evaluation blocks (the env and carry persist)."""
        t = self.torch
        if self._loop_state is None:
            batch = self._alloc_batch()
            metrics_dev = t.zeros(self.nr_epochs * self.nr_minibatches, 10, device=self.device)
            state, _ = self.train_env.reset()
            self._loop_state = [batch, metrics_dev, state.contiguous(), 0, 0, None]
        batch, metrics_dev, state, nr_updates, nr_episodes, prev_end = self._loop_state
        n_upd = self.nr_epochs * self.nr_minibatches
        ev = [t.cuda.Event(enable_timing=True) for _ in range(4)]
        while global_step < self.total_timesteps:
            lr_now = float(self.lr_schedule()[0])
            ev[0].record()
            state = self.collect_rollout(batch, state)
            ev[1].record()
            self.compute_advantages(batch)
            ev[2].record()
            self.update(batch, metrics_dev)
            ev[3].record()
            global_step += self.batch_size
            nr_updates += n_upd
            host = self.reduce_metrics(batch, metrics_dev)     # two library launches + one D2H; raises on a non-finite value
            metrics = {METRIC_NAMES[i]: host[i] for i in (0, 1, 2, 3, 4, 8, 9)}
            metrics["lr/learning_rate"] = lr_now
            metrics["v_value/explained_variance"] = host[10]
            metrics["policy/std_dev"] = host[11]
            metrics["time/acting_time"] = ev[0].elapsed_time(ev[1]) / 1e3
            metrics["time/calc_adv_and_return_time"] = ev[1].elapsed_time(ev[2]) / 1e3
            metrics["time/optimizing_time"] = ev[2].elapsed_time(ev[3]) / 1e3
            if hasattr(self.train_env, "pop_episode_stats"):
                n_done, mean_ret, mean_len = self.train_env.pop_episode_stats()
                nr_episodes += n_done
                if n_done:
                    metrics["rollout/episode_return"] = mean_ret
                    metrics["rollout/episode_length"] = mean_len
            now = time.time()
            if prev_end:
                metrics["time/sps"] = int(self.batch_size / (now - prev_end))
            prev_end = now
            metrics["steps/nr_env_steps"] = global_step
            metrics["steps/nr_updates"] = nr_updates
            metrics["steps/nr_episodes"] = nr_episodes
            self.sink.write(global_step, metrics)
            self.last_metrics = metrics
        self._loop_state = [batch, metrics_dev, state, nr_updates, nr_episodes, prev_end]

    def lr_schedule(self):
        """linear_schedule (ppo_lstm.py:83-85) for the next E*M optimizer steps."""
        n = self.nr_epochs * self.nr_minibatches
        if not self.anneal_learning_rate:
            return np.full(n, self.learning_rate, dtype=np.float32)
        counts = self.opt_count + np.arange(n)
        frac = 1.0 - (counts // n) / max(self.nr_updates, 1)
        return (self.learning_rate * frac).astype(np.float32)

    # ------------------------------------------------------------------ evaluation (ppo_lstm.py:300-334) / test
    def _rollout_deterministic(self, env, nr_steps):
        """Deterministic episodes with a fresh carry.  When the eval env IS the train env (copy_train_env_for_eval) its
        state is saved and put back, so the training episodes, their counter-RNG step and the training carry (which
        belongs to exactly that env state) are untouched -- the reference's evaluation never perturbs training."""
        shared = env is self.train_env
        if shared and not hasattr(env, "snapshot"):
            raise ValueError("ppo_lstm.hip: evaluation on the training env needs env.snapshot()/restore(); "
                             "set environment.copy_train_env_for_eval=False")
        snap = env.snapshot() if shared else None
        try:
            return self._rollout_deterministic_on(env, nr_steps)
        finally:
            if shared:
                env.restore(snap)

    def _rollout_deterministic_on(self, env, nr_steps):
        t = self.torch
        state, _ = env.reset()
        N, A, H = state.shape[0], self.act_dim, self.lstm_hidden       # (a sharded env: this rank's envs)
        f = dict(device=self.device, dtype=t.float32)
        c, h = t.zeros(N, H, **f), t.zeros(N, H, **f)
        action, proc, value, logp = t.empty(N, A, **f), t.empty(N, A, **f), t.empty(N, **f), t.empty(N, **f)
        ep_ret, ep_len = t.zeros(N, **f), t.zeros(N, **f)
        returns, lengths = [], []
        if self.obs_select:
            pobs, cobs = t.empty(N, self.policy_obs_dim, **f), t.empty(N, self.critic_obs_dim, **f)
        for _ in range(nr_steps):
            state, critic_obs = state.contiguous(), None
            if self.obs_select:
                critic_obs = self.ctx.select_columns(state, self.cidx, cobs)
                state = self.ctx.select_columns(state, self.pidx, pobs)
            self.ctx.ppo_lstm_act(self.ldesc, self.pparams, self.cdesc, self.cparams, state, c, h, self.key,
                                  action, proc, value, logp, clip_and_rescale=self.action_clipping_and_rescaling,
                                  act_low=self.act_low, act_high=self.act_high, scheme=self.scheme, deterministic=True,
                                  critic_obs=critic_obs)
            state, reward, terminated, truncated, info = env.step(proc)
            done = terminated | truncated
            self.ctx.lstm_mask_carry(c, h, done.float())
            ep_ret += reward
            ep_len += 1
            if bool(done.any()):
                returns.extend(ep_ret[done].cpu().tolist())
                lengths.extend(ep_len[done].cpu().tolist())
                ep_ret = t.where(done, t.zeros_like(ep_ret), ep_ret)
                ep_len = t.where(done, t.zeros_like(ep_len), ep_len)
        return returns, lengths

    def _evaluate(self, global_step):
        horizon = int(self.horizon or getattr(self.eval_env, "horizon", 1000))
        returns, lengths = self._rollout_deterministic(self.eval_env, horizon)
        self.last_eval = ({"eval/episode_return": float(np.mean(returns)), "eval/episode_length": float(np.mean(lengths))}
                          if returns else {})
        self.sink.write(global_step, self.last_eval)

    def test(self, episodes):
        self.set_eval_mode()
        returns = []
        horizon = int(self.horizon or getattr(self.eval_env, "horizon", 1000))
        while len(returns) < episodes:
            r, _ = self._rollout_deterministic(self.eval_env, horizon)
            returns.extend(r)
        for i, r in enumerate(returns[:episodes]):
            rlx_logger.info(f"Episode {i + 1} - Return: {r}")
        return returns[:episodes]

    # ------------------------------------------------------------------ checkpoint (native format)
    def load(config, train_env, eval_env, run_path, writer, explicitly_set_algorithm_params):
        ckpt = np.load(config.runner.load_model, allow_pickle=False)
        adopt_checkpoint_config(config, json.loads(str(ckpt["config_algorithm"])), explicitly_set_algorithm_params)
        model = PPO_LSTM._load_class(config)(config, train_env, eval_env, run_path, writer)
        for k in ("pparams", "pm", "pv", "cparams", "cm", "cv"):
            getattr(model, k).copy_(model.torch.from_numpy(ckpt[k]).to(model.device))
        model.opt_count = int(ckpt["opt_count"])
        return model

    @staticmethod
    def _load_class(config):
        if str(config.algorithm.name).startswith("ppo_gru"):
            from rlx_amd.algorithms.ppo_gru.hip.ppo_gru import PPO_GRU
            return PPO_GRU
        return PPO_LSTM

    def general_properties():
        return GeneralProperties
