"""sac.hip -- SAC whose update runs as hand-written gfx950 kernels (rl-x_amd/csrc/sac.hip).

Host loop = rl_x/algorithms/sac/flax/sac.py:118-375: per vector step {act (uniform warm-up before
`learning_starts`, then tanh-Gaussian policy) -> env.step -> replay add -> (after learning_starts) sample
`batch_size` transitions -> one `update`}.  The replay ring lives in HBM ([capacity, nr_envs, .] like the
fully-jitted variant, rl_x/algorithms/sac/flax_full_jit/sac.py:139-154); the index draws stay on the host
with numpy's Generator exactly as the reference (`np.random.default_rng(seed)`, sac.py:59,
replay_buffer.py:31-32) so they are bit-reproducible.  PRNG key schedule: K, policy_key,
critic_key, entropy_key = split(PRNGKey(seed), 4); acting K, sub = split(K); update keys = split(K, 2B+1).
"""
import json
import logging
import os
import time

import numpy as np

from rlx_amd.algorithms.sac.hip.general_properties import GeneralProperties
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.plugin import MetricSink, adopt_checkpoint_config

rlx_logger = logging.getLogger("rl_x")

METRIC_NAMES = ["loss/q_loss", "loss/policy_loss", "loss/entropy_loss", "entropy/entropy", "entropy/alpha",
                "q_value/q_value", "gradients/policy_grad_norm", "gradients/critic_grad_norm",
                "gradients/entropy_grad_norm"]


def _lecun_flat(rng, in_dim, hidden, out_dim, ln_first=False):
    """flax Dense default init: lecun_normal kernels, zero biases (sac/flax/policy.py:32-39, critic.py:25-30); LayerNorm
    after the first layer: scale 1, bias 0 (sac/flax_full_jit/policy.py:32-33)."""
    parts, d = [], in_dim
    for li, h in enumerate(list(hidden) + [out_dim]):
        std = np.sqrt(1.0 / d) / 0.87962566103423978
        parts.append((np.clip(rng.standard_normal((d, h)), -2, 2) * std).ravel())
        parts.append(np.zeros(h))
        if ln_first and li == 0:
            parts += [np.ones(h), np.zeros(h)]
        d = h
    return np.concatenate(parts).astype(np.float32)


class SAC:
    def __init__(self, config, train_env, eval_env, run_path, writer):
        import torch
        from rlx_amd.hip import ACT_ELU, ACT_RELU, Ctx, SacHparams, mlp_desc
        from rlx_amd.hip import lib as hiplib
        self.torch, self.hiplib = torch, hiplib
        self.config, self.train_env, self.eval_env, self.writer = config, train_env, eval_env, writer
        self.save_model = config.runner.save_model
        self.save_path = os.path.join(run_path, "models")
        self.track_console = config.runner.track_console
        self.track_tb = config.runner.track_tb
        self.track_wandb = config.runner.track_wandb
        self.seed = config.environment.seed
        self.total_timesteps = config.algorithm.total_timesteps
        self.nr_envs = int(config.environment.nr_envs)
        self.learning_rate = config.algorithm.learning_rate
        self.anneal_learning_rate = config.algorithm.anneal_learning_rate
        self.buffer_size = int(config.algorithm.buffer_size)
        self.learning_starts = config.algorithm.learning_starts
        self.batch_size = int(config.algorithm.batch_size)
        self.tau = config.algorithm.tau
        self.gamma = config.algorithm.gamma
        self.target_entropy = config.algorithm.target_entropy
        self.log_std_min = float(config.algorithm.log_std_min)
        self.log_std_max = float(config.algorithm.log_std_max)
        self.nr_hidden_units = int(config.algorithm.nr_hidden_units)
        self.logging_frequency = config.algorithm.logging_frequency
        self.evaluation_frequency = config.algorithm.evaluation_frequency
        self.evaluation_episodes = int(config.algorithm.evaluation_episodes)
        if self.evaluation_frequency != -1 and self.evaluation_frequency % self.nr_envs != 0:
            raise ValueError("Evaluation frequency must be a multiple of the number of environments.")   # sac.py:72-73
        self.scheme = 1 if config.algorithm.threefry_partitionable else 0
        if config.algorithm.device != "gpu":
            raise ValueError("sac.hip runs on MI355X only: --algorithm.device must be 'gpu' (no CPU fallback)")
        if train_env.general_properties.data_interface_type != DataInterfaceType.TORCH:
            raise ValueError("sac.hip needs a TORCH data-interface environment")
        # Data parallel (one process per GPU): envs and the replay ring are sharded by env column, every rank
        # samples batch_size / world transitions from ITS columns (uniform over the rank's ring: which indices are drawn
        # differs from a one-device draw, the distribution does not), parameters / Adam moments / key are replicated and the
        # library all-reduces [gradients | loss sums] once per update (rlx_sac_hparams.batch_global).
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                self.dist = dist
        except Exception:
            pass
        self.nr_envs_local = int(getattr(train_env, "nr_envs", self.nr_envs // self.world))
        self.env_id_offset = int(getattr(train_env, "env_id_offset", self.rank * self.nr_envs_local))
        if self.nr_envs_local * self.world != self.nr_envs:
            raise ValueError("environment shard size * world size != environment.nr_envs")
        if self.batch_size % self.world != 0:
            raise ValueError("algorithm.batch_size must be divisible by the number of ranks")
        self.batch_local = self.batch_size // self.world
        # (enable_observation_normalization with world > 1: rlx_obs_norm_update_f32 all-reduces the batch sums of a context with a
        #  communicator, so every rank merges the same global batch and the statistics stay replicated)

        self.device = torch.device("cuda", torch.cuda.current_device())
        from rlx_amd.algorithms.ppo.hip.ppo import PPO as _PPO_ctx
        self.ctx = _PPO_ctx._make_ctx(self, Ctx)                        # RCCL communicator in the context (or the gloo hook of the tests)
        # rlx_sac_hparams.keep_images: the networks' split weight images persist between calls and the optimizer kernel keeps them
        # current.  The parameter vectors of this plugin only change through rlx_sac_update_f32; whoever writes them from outside
        # (load() below, a test) calls parameters_written().
        self.keep_images = 1
        # The update gathers the sampled transitions from the ring itself.  batch_states = False (the default: train() never
        # reads them -- the reference's jitted update consumes its batch internally, sac/flax/sac.py:128-215): observation rows wider
        # than 32 columns are left out of the gather (12 of its 44 MB at configs[3]) and self.batch[0:2] stay ZERO; True: the
        # gathered observation rows come back in self.batch[0:2] (tests that compare the sampled batch with the numpy ring).
        self.batch_states = False
        self.sink = MetricSink(rlx_logger, writer, console=self.track_console, tensorboard=self.track_tb, wandb=self.track_wandb,
                               rank=self.rank)
        self.rng = np.random.default_rng(self.seed if self.world == 1 else [int(self.seed), self.rank])   # sac.py:59 (one rank: the reference's stream)
        self.key = hiplib.prng_key(self.seed)                           # sac.py:60-61
        ks = hiplib.threefry_split(self.key, 4, self.scheme)
        self.key, policy_key, critic_key = ks[0], ks[1], ks[2]

        O = int(np.prod(train_env.single_observation_space.shape))
        A = int(np.prod(train_env.single_action_space.shape))
        self.obs_dim, self.act_dim = O, A
        # Networks on a SUBSET of the observation (`x[..., self.policy_observation_indices]`, sac/flax/policy.py:14,31; the
        # critics: critic.py:11,23): the replay ring keeps the env's full rows; the columns are selected from the sampled batch
        # (and from the acting observation) with rlx_select_columns_f32.
        from rlx_amd.algorithms.ppo.hip.ppo import PPO as _PPO
        pidx, cidx = _PPO._observation_indices(train_env, O)
        self.obs_select = pidx is not None
        self.policy_obs_dim, self.critic_obs_dim = (len(pidx), len(cidx)) if self.obs_select else (O, O)
        if self.obs_select:
            self.pidx = torch.from_numpy(pidx).to(torch.device("cuda", torch.cuda.current_device()))
            self.cidx = torch.from_numpy(cidx).to(self.pidx.device)
        # FastSAC's running observation normaliser (fastsac/pytorch/observation_normalizer.py; flag of the same name,
        # fastsac/pytorch/default_config.py): statistics over the env's FULL observation row, updated on the sampled states and
        # next states of every update (fastsac.py:310-311), applied without update when acting / evaluating (:278, :368).
        self.obs_norm = bool(config.algorithm.get("enable_observation_normalization", False))
        if self.obs_norm:
            self.norm_mean, self.norm_var, self.norm_std = (torch.zeros(O, device=self.device), torch.ones(O, device=self.device),
                                                            torch.ones(O, device=self.device))
            self.norm_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.env_as_low = torch.from_numpy(np.asarray(train_env.single_action_space.low, np.float32).reshape(-1)).to(self.device)
        self.env_as_high = torch.from_numpy(np.asarray(train_env.single_action_space.high, np.float32).reshape(-1)).to(self.device)
        if self.target_entropy == "auto":
            self.target_entropy = -float(A)                             # sac.py:69-70
        else:
            self.target_entropy = float(self.target_entropy)
        H = self.nr_hidden_units
        arch = config.algorithm.get("network_architecture", "flax")
        if arch == "full_jit":                 # sac/flax_full_jit/policy.py:29-41, critic.py:20-31
            hidden, act, ln = [512, 256, 128], ACT_ELU, True
        elif arch == "flax":                   # sac/flax/policy.py:22-41, critic.py:17-53
            hidden, act, ln = [H, H], ACT_RELU, False
        else:
            raise ValueError("algorithm.network_architecture must be 'flax' or 'full_jit'")
        self.full_jit = arch == "full_jit"     # also selects the key schedule and the device-side replay index draw
        Op, Oc = self.policy_obs_dim, self.critic_obs_dim
        self.pdesc = mlp_desc(Op, hidden, 2 * A, act, ln, False)
        self.qdesc = mlp_desc(Oc + A, hidden, 1, act, ln, False)
        prng = np.random.default_rng([int(policy_key[0]), int(policy_key[1])])
        crng = np.random.default_rng([int(critic_key[0]), int(critic_key[1])])
        dev = self.device
        self.pparams = torch.from_numpy(_lecun_flat(prng, Op, hidden, 2 * A, ln)).to(dev)
        q = np.concatenate([_lecun_flat(crng, Oc + A, hidden, 1, ln) for _ in range(2)])
        self.qparams = torch.from_numpy(q).to(dev)
        self.qtarget = self.qparams.clone()                             # target = same init
        self.log_alpha = torch.zeros(1, device=dev)                     # EntropyCoefficient(1.0): log(1.0)
        self.pm, self.pv = torch.zeros_like(self.pparams), torch.zeros_like(self.pparams)
        self.qm, self.qv = torch.zeros_like(self.qparams), torch.zeros_like(self.qparams)
        self.am, self.av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        self.opt_count = 0
        self.SacHparams = SacHparams
        if self.save_model:
            os.makedirs(self.save_path, exist_ok=True)
            self.best_mean_return = -np.inf
            self.best_model_file_name = "best.model"

    def current_lr(self):
        if not self.anneal_learning_rate:                               # sac.py:79-83
            return self.learning_rate
        step = self.opt_count * self.nr_envs - self.learning_starts
        return self.learning_rate * (1.0 - step / (self.total_timesteps - self.learning_starts))

    def hparams(self):
        lr = self.current_lr()
        hp = self.SacHparams(self.gamma, self.tau, self.target_entropy, self.log_std_min, self.log_std_max, lr, lr,
                             lr, 0.9, 0.999, 1e-8, int(self.full_jit))
        hp.keep_images = int(getattr(self, "keep_images", 0))
        return hp

    def parameters_written(self):
        """pparams / qparams / qtarget were written from outside rlx_sac_update_f32: drop the kept weight images."""
        self.ctx.sac_invalidate_images()

    def processed_action(self, action):                                 # sac/flax/policy.py:44-48
        return self.env_as_low + 0.5 * (action.clamp(-1, 1) + 1.0) * (self.env_as_high - self.env_as_low)

    def warmup_actions(self, gen):
        """Uniform actions in [-1, 1) of the steps before `learning_starts` (sac/flax/sac.py:251-253).  The host-loop flavour
        samples them outside the JAX key stream (numpy / the torch generator here).  The fully jitted flavour draws them FROM
        the key -- `key, subkey = split(key)` per prefill step, `action_space.sample(subkey)`, sac/flax_full_jit/sac.py:160-164
        -- so the key entering the first update has advanced by one split per prefill step: reproduced here (one uniform draw
        per env and action dimension from the subkey's threefry bits)."""
        t = self.torch
        nl, off = getattr(self, "nr_envs_local", self.nr_envs), getattr(self, "env_id_offset", 0)
        if not getattr(self, "full_jit", False):
            return t.rand(nl, self.act_dim, device=self.device, generator=gen) * 2.0 - 1.0
        ks = self.hiplib.threefry_split(self.key, 2, self.scheme)
        self.key, sub = ks[0], ks[1]
        bits = t.empty(self.nr_envs * self.act_dim, dtype=t.int32, device=self.device)   # the draw for ALL envs; this rank's rows
        self.ctx.random_bits(sub, bits, self.scheme)
        unit = (((bits >> 9) & 0x7FFFFF) | 0x3F800000).view(t.float32) - 1.0          # jax.random.uniform: mantissa bits -> [0, 1)
        return (unit * 2.0 - 1.0).view(self.nr_envs, self.act_dim)[off:off + nl]

    def policy_obs(self, state):
        """The observation columns the policy reads (the whole row unless the env defines policy_observation_indices)."""
        if getattr(self, "obs_norm", False):                            # normalise the full row first, statistics frozen
            if getattr(self, "act_norm", None) is None or self.act_norm.shape[0] != state.shape[0]:
                self.act_norm = self.torch.empty(state.shape[0], self.obs_dim, device=self.device)
            state = self.ctx.obs_norm_apply(state.contiguous(), self.norm_mean, self.norm_std, self.act_norm)
        if not getattr(self, "obs_select", False):
            return state
        if getattr(self, "act_obs", None) is None or self.act_obs.shape[0] != state.shape[0]:
            self.act_obs = self.torch.empty(state.shape[0], self.policy_obs_dim, device=self.device)
        return self.ctx.select_columns(state.contiguous(), self.pidx, self.act_obs)

    def _alloc(self):
        t = self.torch
        N, O, A, B = getattr(self, "nr_envs_local", self.nr_envs), self.obs_dim, self.act_dim, getattr(self, "batch_local", self.batch_size)
        cap = self.buffer_size // self.nr_envs                          # replay_buffer.py:8 (rows; sharded: N_local columns each)
        f = dict(device=self.device, dtype=t.float32)
        self.ring = (t.zeros(cap, N, O, **f), t.zeros(cap, N, O, **f), t.zeros(cap, N, A, **f), t.zeros(cap, N, **f),
                     t.zeros(cap, N, **f))
        self.capacity, self.pos, self.size = cap, 0, 0
        self.batch = (t.zeros(B, O, **f), t.zeros(B, O, **f), t.zeros(B, A, **f), t.zeros(B, **f), t.zeros(B, **f))
        self.idx1 = t.empty(B, dtype=t.int32, device=self.device)
        self.idx2 = t.empty(B, dtype=t.int32, device=self.device)
        if getattr(self, "world", 1) > 1:        # full-jit flavour: the global draw, of which this rank takes its slice
            self.idx_g = (t.empty(self.batch_size, dtype=t.int32, device=self.device), t.empty(self.batch_size, dtype=t.int32, device=self.device))
        self.action = t.empty(N, A, **f)
        self.metrics_dev = t.zeros(10, **f)
        if getattr(self, "obs_select", False):   # selected columns of the sampled batch (policy: s, s'; critics: s, s') and of the acting observation
            Op, Oc = self.policy_obs_dim, self.critic_obs_dim
            self.sel = (t.empty(B, Op, **f), t.empty(B, Op, **f), t.empty(B, Oc, **f), t.empty(B, Oc, **f))
            self.act_obs = t.empty(N, Op, **f)

    def replay_add(self, state, next_state, action, reward, terminated):
        for dst, src in zip(self.ring, (state, next_state, action, reward, terminated)):
            dst[self.pos].copy_(src)
        self.pos = (self.pos + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def _host_indices(self, bl, nl, ahead=32):
        """The reference's index draws -- numpy PCG64, `rng.integers(size, B)` then `rng.integers(nr_envs, B)` per update
        (replay_buffer.py:31-32) -- for a block of upcoming updates at once, in ONE pinned H2D copy instead of two copies per step.
        The stream stays the reference's bit for bit: the draw of update j uses the ring size that update will see, the
        generator state in front of every update's draw is kept, and whenever the size an update actually finds differs from the
        predicted one the generator is rewound to that update and the block is redrawn from there.

        Prediction: the ring grows by the stride observed between the last two updates (one row per update in the usual
        step / update cadence, zero with several updates per step or a full ring).  A miss halves the block length (down to one
        update, i.e. no speculation) and a block consumed without a miss doubles it again up to `ahead`, so a cadence the
        predictor cannot follow costs one redraw per update, not `ahead` of them.

        The returned tensors are VIEWS into the block's device buffer; the next block's asynchronous copy overwrites it.  That
        is safe because the copy and rlx_sac_replay_sample (the only reader) are both issued on torch's current stream -- do
        not move either to another stream.  self.rng runs ahead of consumption; checkpoint `consumed_rng_state()` instead."""
        t = self.torch
        c = getattr(self, "_idx_cache", None)
        if c is None or c["bl"] != bl or c["cap"] != ahead:
            c = self._idx_cache = dict(bl=bl, cap=ahead, len=ahead, pos=0, n=0, sizes=[], states=[], last=None, stride=1, clean=True,
                                       host=t.empty(ahead, 2, bl, dtype=t.int32).pin_memory(),
                                       dev=t.empty(ahead, 2, bl, dtype=t.int32, device=self.device), ev=t.cuda.Event())
        if c["last"] is not None:
            c["stride"] = max(self.size - c["last"], 0)
        c["last"] = self.size
        if c["pos"] < c["n"] and c["sizes"][c["pos"]] != self.size:      # prediction missed: back to the state before this draw
            self.rng.bit_generator.state = c["states"][c["pos"]]
            c["n"] = c["pos"] = 0
            c["len"], c["clean"] = max(c["len"] // 2, 1), False
        elif c["pos"] == c["n"] and c["n"]:
            c["len"] = min(2 * c["len"], c["cap"]) if c["clean"] else c["len"]
        if c["pos"] == c["n"]:
            c["ev"].synchronize()                                          # the previous block's copy has left the pinned buffer
            host, n = c["host"].numpy(), c["len"]
            c["sizes"] = [min(self.size + j * c["stride"], self.capacity) for j in range(n)]
            c["states"] = []
            for j, sz in enumerate(c["sizes"]):
                c["states"].append(self.rng.bit_generator.state)
                host[j, 0] = self.rng.integers(sz, size=bl)
                host[j, 1] = self.rng.integers(nl, size=bl)
            c["dev"][:n].copy_(c["host"][:n], non_blocking=True)
            c["ev"].record()
            c["pos"], c["n"], c["clean"] = 0, n, True
        j = c["pos"]
        c["pos"] += 1
        return c["dev"][j, 0], c["dev"][j, 1]

    def consumed_rng_state(self):
        """The numpy generator state as the reference's would be after the updates done so far (self.rng itself has already
        drawn the rest of the current block)."""
        c = getattr(self, "_idx_cache", None)
        if c is None or c["pos"] >= c["n"]:
            return self.rng.bit_generator.state
        return c["states"][c["pos"]]

    def sample_and_update(self):
        t = self.torch
        world, nl, bl = getattr(self, "world", 1), getattr(self, "nr_envs_local", self.nr_envs), getattr(self, "batch_local", self.batch_size)
        if self.full_jit and world > 1:   # the global draw (same on every rank); rows [rank * bl, ..) of it, env index folded into the shard
            g1, g2 = self.idx_g
            self.ctx.sac_replay_draw(self.key, self.batch_size, self.size, self.nr_envs, g1, g2, self.scheme)
            self.idx1.copy_(g1[self.rank * bl:(self.rank + 1) * bl])
            t.remainder(g2[self.rank * bl:(self.rank + 1) * bl], nl, out=self.idx2)
        elif self.full_jit:   # sac/flax_full_jit/sac.py:273-282: indices from keys[1] of this update's split, on the device
            self.ctx.sac_replay_draw(self.key, self.batch_size, self.size, self.nr_envs, self.idx1, self.idx2, self.scheme)
        else:
            self.idx1, self.idx2 = self._host_indices(bl, nl)
        batch, hp = self.batch, self.hparams()
        if getattr(self, "obs_norm", False) or getattr(self, "obs_select", False):
            self.ctx.sac_replay_sample(self.ring, self.idx1, self.idx2, self.batch)    # the batch is transformed before the update
        else:   # the update gathers the transitions itself (into self.batch), in the launch that lays out the critics' input rows
            (hp.ring_states, hp.ring_next_states, hp.ring_actions, hp.ring_rewards,
             hp.ring_terminations) = (x.data_ptr() for x in self.ring)
            hp.ring_idx1, hp.ring_idx2, hp.ring_nr_envs = self.idx1.data_ptr(), self.idx2.data_ptr(), self.ring[0].shape[1]
            if not getattr(self, "batch_states", True) and self.obs_dim > 32:      # rlx_sac_update_f32: states / next_states = NULL
                batch = (None, None) + tuple(batch[2:])
        if getattr(self, "obs_norm", False):     # states first, then next states, each updating the statistics (fastsac.py:310-311)
            for x in batch[:2]:
                self.ctx.obs_norm_update(x, self.norm_mean, self.norm_var, self.norm_std, self.norm_count)
                self.ctx.obs_norm_apply(x, self.norm_mean, self.norm_std, x)
        if getattr(self, "obs_select", False):
            sp, s2p, sc, s2c = self.sel
            self.ctx.select_columns(batch[0], self.pidx, sp)
            self.ctx.select_columns(batch[1], self.pidx, s2p)
            self.ctx.select_columns(batch[0], self.cidx, sc)
            self.ctx.select_columns(batch[1], self.cidx, s2c)
            batch = (sp, s2p) + tuple(batch[2:])
            hp.critic_states, hp.critic_next_states = sc.data_ptr(), s2c.data_ptr()
        if world > 1:
            hp.batch_global, hp.batch_row_offset = self.batch_size, self.rank * bl
        self.key, self.opt_count = self.ctx.sac_update(
            self.pdesc, self.pparams, self.pm, self.pv, self.qdesc, self.qparams, self.qm, self.qv, self.qtarget,
            self.log_alpha, self.am, self.av, batch, self.key, self.opt_count, hp, self.metrics_dev, self.scheme)

    def vector_step(self, env, state, warmup=False, gen=None):
        """act -> env.step -> replay add; returns the next observation.
        Device envs that can write a transition into caller-provided rows (`step_into`): the replay ring slot IS the
        destination of the acting kernel (action) and of the env kernel (pre-step observation, final observation, reward,
        termination) -- no copy per step instead of five, no mask / cast / clone kernels."""
        t = self.torch
        if getattr(self, "_half_range", None) is None:
            self._half_range = (0.5 * (self.env_as_high - self.env_as_low)).contiguous()
            self._low = self.env_as_low.contiguous()
            self._processed = t.empty(getattr(self, "nr_envs_local", self.nr_envs), self.act_dim, device=self.device)
        # low + 0.5 * (clip(a, -1, 1) + 1) * (high - low)  (sac/flax/policy.py:44-48; the factor 0.5 commutes exactly): policy
        # actions get it from the acting launch itself, the uniform warm-up actions from three torch kernels
        fused = (self._low, self._half_range, self._processed)
        if getattr(self, "direct_replay", True) and hasattr(env, "step_into") and hasattr(env, "obs"):
            ring_s, ring_ns, ring_a, ring_r, ring_t = (x[self.pos] for x in self.ring)
            # the pre-step observation row: stored by the env kernel itself when the env offers it, else one copy here
            prev_in_step = getattr(env, "step_into_prev_obs", False)
            if not prev_in_step:
                ring_s.copy_(env.obs)
            if warmup:
                ring_a.copy_(self.warmup_actions(gen))
                t.addcmul(self._low, t.clamp(ring_a, -1.0, 1.0).add_(1.0), self._half_range, out=self._processed)
            else:
                self.key = self.ctx.sac_act(self.pdesc, self.pparams, self.policy_obs(env.obs), self.key, ring_a,
                                            self.log_std_min, self.log_std_max, scheme=self.scheme, processed=fused,
                                            row_offset=getattr(self, "env_id_offset", 0), n_global=self.nr_envs)
            if prev_in_step:
                env.step_into(self._processed, ring_ns, ring_r, ring_t, prev_obs_out=ring_s)
            else:
                env.step_into(self._processed, ring_ns, ring_r, ring_t)
            self.pos = (self.pos + 1) % self.capacity
            self.size = min(self.size + 1, self.capacity)
            return env.obs
        if warmup:
            action = self.warmup_actions(gen)
            t.addcmul(self._low, t.clamp(action, -1.0, 1.0).add_(1.0), self._half_range, out=self._processed)
        else:
            self.key = self.ctx.sac_act(self.pdesc, self.pparams, self.policy_obs(state), self.key, self.action,
                                        self.log_std_min, self.log_std_max, scheme=self.scheme, processed=fused,
                                            row_offset=getattr(self, "env_id_offset", 0), n_global=self.nr_envs)
            action = self.action
        next_state, reward, terminated, truncated, info = env.step(self._processed)
        fin = info.get("final_observation") if isinstance(info, dict) else None
        self.replay_add(state, fin if fin is not None else next_state, action, reward, terminated.float())
        return next_state.clone()

    def train(self):
        t = self.torch
        self._alloc()
        env = self.train_env
        state, _ = env.reset()
        state = state.clone()
        global_step, nr_updates, nr_episodes = 0, 0, 0
        metric_sum = t.zeros(10, device=self.device)
        metric_n = 0
        last_log_time, last_log_step = time.time(), 0
        gen = t.Generator(device=self.device)
        gen.manual_seed(int(self.seed) + getattr(self, "rank", 0))
        pending_eval = {}
        while global_step < self.total_timesteps:
            state = self.vector_step(env, state, global_step < self.learning_starts, gen)   # sac.py:251-253: uniform warm-up
            global_step += self.nr_envs
            if global_step > self.learning_starts:
                self.sample_and_update()
                metric_sum += self.metrics_dev
                metric_n += 1
                nr_updates += 1
            # Evaluating (sac.py:303-321): deterministic tanh(mean) actions until evaluation_episodes episodes finished
            if self.evaluation_frequency != -1 and global_step % self.evaluation_frequency == 0:
                eval_returns, eval_lengths = self.evaluate(self.evaluation_episodes)
                self.last_eval = {"eval/episode_return": float(np.mean(eval_returns)),
                                  "eval/episode_length": float(np.mean(eval_lengths))}
                pending_eval = dict(self.last_eval)
            if global_step % self.logging_frequency < self.nr_envs or global_step >= self.total_timesteps:
                now = time.time()
                m = (metric_sum / max(metric_n, 1)).cpu().tolist()     # ONE D2H per logging interval
                if metric_n and not all(np.isfinite(v) for v in m[:9]):
                    raise FloatingPointError(
                        "sac.hip: non-finite loss / gradient norm since the last log " + str([round(v, 6) for v in m[:9]]) +
                        ".  If the training itself is sane, an operand left the fp16 window of the split-operand GEMM engine "
                        "(|activation or observation| >= 4094, |weight| >= 1023; rl-x_amd/csrc/gemm_bx.h): rerun with "
                        "RLX_GEMM_BX=0 (exact-fp32 MFMA engine) or enable observation normalisation.")
                combined = {METRIC_NAMES[i]: m[i] for i in range(9)} if metric_n else {}
                if hasattr(env, "pop_episode_stats"):
                    n_done, mean_ret, mean_len = env.pop_episode_stats()
                    nr_episodes += n_done
                    if n_done:
                        combined.update({"rollout/episode_return": mean_ret, "rollout/episode_length": mean_len})
                        # sac.py:302-309: keep the best model by mean episode return once learning has started
                        if self.save_model and getattr(self, "rank", 0) == 0 and global_step > self.learning_starts and mean_ret > self.best_mean_return:
                            self.best_mean_return = mean_ret
                            self.save()
                combined.update(pending_eval)
                pending_eval = {}
                combined.update({"steps/nr_env_steps": global_step, "steps/nr_updates": nr_updates,
                                 "steps/nr_episodes": nr_episodes, "lr/learning_rate": self.current_lr(),
                                 "time/sps": int((global_step - last_log_step) / max(now - last_log_time, 1e-9))})
                last_log_time, last_log_step = now, global_step
                metric_sum.zero_()
                metric_n = 0
                self.sink.write(global_step, combined)
                self.last_metrics = combined

    def evaluate(self, episodes):
        """Deterministic episodes (tanh(mean), sac.py:217-221) on the eval env -> (returns, lengths).  A shared train / eval
        env (copy_train_env_for_eval) is snapshotted and restored: the reference's evaluation never perturbs training."""
        env = self.eval_env
        shared = env is self.train_env
        if shared and not hasattr(env, "snapshot"):
            raise ValueError("sac.hip: evaluation on the training env needs env.snapshot()/restore(); "
                             "set environment.copy_train_env_for_eval=False")
        snap = env.snapshot() if shared else None
        try:
            t = self.torch
            state, _ = env.reset()
            ne = state.shape[0]
            action = t.empty(ne, self.act_dim, device=self.device)
            returns, lengths = [], []
            ep_ret, ep_len = t.zeros(ne, device=self.device), t.zeros(ne, device=self.device)
            while len(returns) < episodes:
                self.ctx.sac_act(self.pdesc, self.pparams, self.policy_obs(state.contiguous()), self.key, action,
                                 self.log_std_min, self.log_std_max, deterministic=True)
                state, reward, terminated, truncated, info = env.step(self.processed_action(action))
                ep_ret += reward
                ep_len += 1
                done = terminated | truncated
                if bool(done.any()):
                    returns.extend(ep_ret[done].cpu().tolist())
                    lengths.extend(ep_len[done].cpu().tolist())
                    ep_ret = t.where(done, t.zeros_like(ep_ret), ep_ret)
                    ep_len = t.where(done, t.zeros_like(ep_len), ep_len)
            return returns[:episodes], lengths[:episodes]
        finally:
            if shared:
                env.restore(snap)

    def test(self, episodes):
        return self.evaluate(episodes)[0]

    _STATE = ("pparams", "pm", "pv", "qparams", "qm", "qv", "qtarget", "log_alpha", "am", "av")
    _NORM_STATE = ("norm_mean", "norm_var", "norm_std", "norm_count")      # observation_normalizer_state_dict (fastsac.py:476)

    def save(self):
        """Native checkpoint: flat parameter / Adam-moment vectors + the algorithm config, one .npz
        (the reference zips an orbax PyTree + config_algorithm.json, sac.py:382-399)."""
        path = os.path.join(self.save_path, self.best_model_file_name)
        state = {k: getattr(self, k).cpu().numpy() for k in self._STATE + (self._NORM_STATE if self.obs_norm else ())}
        np.savez(path + ".tmp.npz", opt_count=self.opt_count, key=self.key,
                 config_algorithm=json.dumps(self.config.algorithm.to_dict()), **state)
        os.replace(path + ".tmp.npz", path)

    def load(config, train_env, eval_env, run_path, writer, explicitly_set_algorithm_params):
        ckpt = np.load(config.runner.load_model, allow_pickle=False)
        adopt_checkpoint_config(config, json.loads(str(ckpt["config_algorithm"])), explicitly_set_algorithm_params)   # sac.py:409-412
        model = SAC(config, train_env, eval_env, run_path, writer)
        for k in SAC._STATE + (SAC._NORM_STATE if model.obs_norm else ()):
            getattr(model, k).copy_(model.torch.from_numpy(ckpt[k]).to(model.device))
        model.parameters_written()
        model.opt_count = int(ckpt["opt_count"])
        model.key = ckpt["key"].astype(np.uint32)
        return model

    def general_properties():
        return GeneralProperties
