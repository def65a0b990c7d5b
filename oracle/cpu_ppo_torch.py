"""CPU baseline (test/bench infrastructure, NOT the product): the same PPO iteration as
rl_x/algorithms/ppo/flax_full_jit/ppo.py:130-265 on torch-CPU fp32 with all host threads --
the stand-in for RL-X's own Flax-CPU path, which cannot run here (JAX / Flax / Optax are not
installable).  Forward/backward by torch.autograd, GEMMs
by torch's multi-threaded CPU BLAS, i.e. what XLA:CPU would also spend its time in.

`time_iteration` warms up, then times a BOUNDED sample (>= 32 acting steps, one full epoch of minibatch
updates, the whole GAE) and scales each phase to one full iteration (T acting steps, E*M updates) --
kind "port" in bench.py's cpu_baseline.
"""
import math
import time

import numpy as np

from . import nets
from .env import RandomObsEnvOracle

LOG_2PI = math.log(2 * math.pi)


def _loss(pspec, pp, cspec, cp, s, a, lp_old, ret, adv, clip, ent_c, v_c):
    import torch
    A = pspec.out_dim
    mean = nets.torch_forward(pspec, pp, s)
    logstd = pp[pspec.logstd:pspec.logstd + A][None, :]
    std = torch.exp(logstd)
    nlp = (-0.5 * ((a - mean) / std) ** 2 - 0.5 * LOG_2PI - logstd).sum(1)
    ent = (logstd + 0.5 * math.log(2 * math.pi * math.e)).sum(1)
    ratio = torch.exp(nlp - lp_old)
    pg = torch.maximum(-adv * ratio, -adv * torch.clamp(ratio, 1 - clip, 1 + clip))
    v = nets.torch_forward(cspec, cp, s).reshape(-1)
    return (pg - ent_c * ent + v_c * 0.5 * (v - ret) ** 2).mean()


def usable_cores():
    """Host cores this process may really use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _gae_vectorised(rewards, values, next_values, terminations, gamma, lam):
    """calculate_gae_advantages (ppo/flax/ppo.py:122-135) with numpy row operations: one pass over T, vectors over N."""
    T = rewards.shape[0]
    delta = rewards + gamma * next_values * (1.0 - terminations) - values
    adv = np.empty_like(delta)
    adv[T - 1] = delta[T - 1]
    decay = gamma * lam * (1.0 - terminations)
    for t in range(T - 2, -1, -1):
        adv[t] = delta[t] + decay[t] * adv[t + 1]
    return adv, adv + values


def time_iteration(N=4096, T=128, O=17, A=6, E=10, mb=32768, arch="B", sample_steps=32, sample_updates=None,
                   threads=None, seed=1, warmup_steps=4, warmup_updates=2):
    """Times a BOUNDED sample of the iteration after a warm-up -- `sample_steps` (>= 32) of the T acting steps, the critic
    pass over next_states on one minibatch-sized slice, the whole GAE (vectorised), and one full epoch of minibatch
    updates (M = B / mb, 16 at the benchmark config) -- and scales each phase to one full iteration (T steps, E epochs).
    Returns dict(env_steps_per_s, seconds_per_iteration, cores, sample)."""
    import torch
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    pspec, cspec = nets.make_spec(arch, O, A, True), nets.make_spec(arch, O, 1, False)
    pp = torch.tensor(nets.init_params(pspec, rng, 0.01), requires_grad=True)
    cp = torch.tensor(nets.init_params(cspec, rng, 1.0), requires_grad=True)
    opt = torch.optim.Adam([pp, cp], lr=4e-4)
    env = RandomObsEnvOracle(seed, N, O, A)
    obs = torch.from_numpy(env.reset())
    B = N * T
    M = B // mb
    sample_updates = sample_updates or M

    def act_step(obs):
        with torch.no_grad():
            mean = nets.torch_forward(pspec, pp, obs)
            logstd = pp[pspec.logstd:pspec.logstd + A][None, :]
            act = mean + torch.exp(logstd) * torch.randn_like(mean)
            _ = (-0.5 * ((act - mean) / torch.exp(logstd)) ** 2 - 0.5 * LOG_2PI - logstd).sum(1)
            _ = nets.torch_forward(cspec, cp, obs)
            nobs, fin, r, term, trunc, done = env.step(act.numpy())
        return torch.from_numpy(nobs)
    # --- acting: warm-up, then the sample
    for _ in range(warmup_steps):
        obs = act_step(obs)
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        obs = act_step(obs)
    t_act = (time.perf_counter() - t0) / sample_steps
    # --- critic on next_states (timed on one minibatch-sized slice, scaled to the batch) + GAE over the whole batch
    states = torch.randn(mb, O)
    with torch.no_grad():
        _ = nets.torch_forward(cspec, cp, states)
        t0 = time.perf_counter()
        _ = nets.torch_forward(cspec, cp, states)
    t_nextv = (time.perf_counter() - t0) * (B / mb)
    r = np.random.default_rng(0).standard_normal((T, N)).astype(np.float32)
    t0 = time.perf_counter()
    _gae_vectorised(r, r, r, np.zeros_like(r), 0.99, 0.9)
    t_gae = time.perf_counter() - t0
    # --- minibatch updates: warm-up, then one epoch
    actions = torch.randn(mb, A)
    lp = torch.randn(mb)
    ret = torch.randn(mb)
    adv = torch.randn(mb)

    def update():
        advn = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
        opt.zero_grad()
        loss = _loss(pspec, pp, cspec, cp, states, actions, lp, ret, advn, 0.1, 0.0, 1.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([pp], 5.0)
        torch.nn.utils.clip_grad_norm_([cp], 5.0)
        opt.step()
    for _ in range(warmup_updates):
        update()
    t0 = time.perf_counter()
    for _ in range(sample_updates):
        update()
    t_upd = (time.perf_counter() - t0) / sample_updates
    sec = T * t_act + t_nextv + t_gae + E * M * t_upd
    return {"env_steps_per_s": B / sec, "seconds_per_iteration": sec, "cores": threads,
            "phases_s": {"acting_step": t_act, "critic_next_states": t_nextv, "gae": t_gae, "minibatch_update": t_upd},
            "sample": f"after a warm-up ({warmup_steps} acting steps, {warmup_updates} updates): {sample_steps} of {T} acting "
                      f"steps + critic(next_states) on {mb} of {B} rows + the whole GAE (vectorised) + {sample_updates} of "
                      f"{E * M} minibatch updates (mb={mb}; one epoch = {M}), each phase scaled to one iteration; "
                      f"torch-CPU fp32, {threads} threads, arch {arch}"}
