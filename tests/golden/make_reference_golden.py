"""Golden vectors produced by EXECUTING the reference's own code (its PyTorch flavour) in the authoring container.

    python tests/golden/make_reference_golden.py          # needs /root/reference; writes tests/golden/reference_*.npz

The reference's JAX flavours (the parity target) cannot run here (no jax/flax/optax), and its PyTorch flavour cannot be
imported as a package either (`import wandb`, `ml_collections` at module scope).  What CAN run with torch + numpy alone:

  * the network modules  rl_x/algorithms/ppo/pytorch/policy.py  (ContinuousFlatValuesPolicy)
                         rl_x/algorithms/ppo/pytorch/critic.py  (FlatValuesCritic)
                         rl_x/algorithms/sac/pytorch/policy.py  (Policy), q_network.py (QNetwork),
                         entropy_coefficient.py (EntropyCoefficient)
    -- loaded by FILE PATH (importlib), so the package __init__ (which pulls in wandb) never runs;
  * the arithmetic closures defined inside `PPO.train` / `SAC.train`
        calculate_gae_advantages_and_returns, policy_loss_fn, critic_loss_fn        (ppo/pytorch/ppo.py:98-166)
        policy_and_entropy_loss_fn, critic_loss_fn                                  (sac/pytorch/sac.py:90-166)
    -- their AST nodes are compiled from the reference file where it lies (decorators `torch.jit.script` /
    `torch.compile` dropped: they do not change the arithmetic) and run against a stand-in `self` that holds the
    reference's modules, `torch.optim.Adam` optimisers constructed as the reference constructs them
    (ppo.py:82-84, sac.py:70-76) and the hyper-parameters.

Nothing of the reference's text is stored: the fixtures hold inputs and the outputs the reference code produced.
The PyTorch flavour shares the algorithm with the JAX flavours except for details the tests account for explicitly
(tests/test_oracle_reference_pin.py): `Tensor.std()` is the unbiased estimator (jnp.std: population),
`clip_grad_norm_` divides by (norm + 1e-6), `critic_loss` is reported times critic_coef, SAC steps critic / policy /
alpha one after the other (the JAX flavour differentiates one combined loss).  Networks: 2 x tanh (PPO, arch "A"),
2 x relu (SAC) -- the same layers the HIP kernels implement; the LayerNorm/ELU torso of flax_full_jit has no PyTorch twin
in the reference and stays pinned by torch.nn.functional only.
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RLX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_by_path(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def train_closures(rel, names, namespace):
    """Compile the nested functions `names` of `<class>.train` from the reference file and bind them to `namespace`
    (their free variable `self` resolves there)."""
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read(), filename=path)
    found = []
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "train"]:
            for node in fn.body:
                if isinstance(node, ast.FunctionDef) and node.name in names:
                    node.decorator_list = []
                    found.append(node)
    assert sorted(n.name for n in found) == sorted(names), [n.name for n in found]
    mod = ast.Module(body=found, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), namespace)
    return [namespace[n] for n in names]


def fake_env(O, A):
    sp = types.SimpleNamespace
    return sp(single_action_space=sp(low=-np.ones(A, np.float32), high=np.ones(A, np.float32), shape=(A,)),
              single_observation_space=sp(shape=(O,)))


def flat_mlp(linears, extra=()):
    """torch Linear stack -> the flat layout of include/rlx_hip.h: per layer W[in,out] row-major, b[out]; then extras."""
    parts = []
    for lin in linears:
        parts += [lin.weight.detach().T.contiguous().reshape(-1), lin.bias.detach().reshape(-1)]
    parts += [e.detach().reshape(-1) for e in extra]
    return torch.cat(parts).numpy().copy()


def flat_grads(linears, extra=()):
    parts = []
    for lin in linears:
        parts += [lin.weight.grad.T.contiguous().reshape(-1), lin.bias.grad.reshape(-1)]
    parts += [e.grad.reshape(-1) for e in extra]
    return torch.cat(parts).numpy().copy()


def jitter(module, gen, scale):
    with torch.no_grad():
        for p in module.parameters():
            p.add_(scale * torch.randn(p.shape, generator=gen, dtype=p.dtype))


def rounder(dtype, round_inputs):
    """round_inputs (the "f64r" fixtures): every array the HIP path receives as an INPUT -- parameters, rollout data and the
    intermediate results the reference feeds into its next stage -- is rounded to the nearest float32 before the reference's
    float64 arithmetic consumes it.  The fp32 kernels then see bit-identical inputs, and the difference to these fixtures is
    their own arithmetic error only (no input-rounding term): the 1e-5 bar is checked against exact-arithmetic results."""
    if not round_inputs:
        return lambda x: x
    return lambda x: x.to(torch.float32).to(dtype)


def round_module(module, R):
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(R(p))


# ------------------------------------------------------------------------------------------------ PPO
def make_ppo(dtype, tag, round_inputs=False):
    R = rounder(dtype, round_inputs)
    sys.path.insert(0, REF)
    pol_mod = load_by_path("rl_x/algorithms/ppo/pytorch/policy.py", "ref_ppo_policy")
    cri_mod = load_by_path("rl_x/algorithms/ppo/pytorch/critic.py", "ref_ppo_critic")
    O, A, H, T, N, MB = 11, 3, 64, 16, 8, 64
    torch.manual_seed(3)
    gen = torch.Generator().manual_seed(5)
    env = fake_env(O, A)
    policy = pol_mod.ContinuousFlatValuesPolicy(env, 0.7, True, H, "cpu", np.arange(O)).to(dtype)
    critic = cri_mod.FlatValuesCritic(env, H, "cpu", np.arange(O)).to(dtype)
    jitter(policy, gen, 0.05)
    jitter(critic, gen, 0.05)
    round_module(policy, R)
    round_module(critic, R)
    plin = [m for m in policy.policy_mean if isinstance(m, torch.nn.Linear)]
    clin = [m for m in critic.critic if isinstance(m, torch.nn.Linear)]
    hp = dict(gamma=0.99, gae_lambda=0.95, clip_range=0.2, entropy_coef=0.01, critic_coef=0.5, max_grad_norm=0.5,
              learning_rate=3e-4)
    self = types.SimpleNamespace(policy=policy, critic=critic, bf16_mixed_precision_training=False, compile_mode="default", **hp)
    # optimisers exactly as ppo/pytorch/ppo.py:82-84 on a CPU device
    self.policy_optimizer = torch.optim.Adam(policy.parameters(), lr=hp["learning_rate"], fused=False)
    self.critic_optimizer = torch.optim.Adam(critic.parameters(), lr=hp["learning_rate"], fused=False)
    ns = {"torch": torch, "nn": torch.nn, "autocast": torch.amp.autocast, "self": self}
    gae_fn, policy_loss_fn, critic_loss_fn = train_closures(
        "rl_x/algorithms/ppo/pytorch/ppo.py", ["calculate_gae_advantages_and_returns", "policy_loss_fn", "critic_loss_fn"], ns)

    r = lambda *s: R(torch.randn(*s, generator=gen, dtype=dtype))
    states, next_states = r(T, N, O), r(T, N, O)
    rewards = r(T, N)
    term = (torch.rand(T, N, generator=gen) < 0.15).to(dtype)
    out = {"obs_dim": O, "act_dim": A, "hidden": H, "source": "reference:rl_x/algorithms/ppo/pytorch (executed)", **hp}
    out["pparams0"] = flat_mlp(plin, [policy.policy_logstd])
    out["cparams0"] = flat_mlp(clin)
    with torch.no_grad():
        action, scaled, logp = policy.get_action_logprob(states.reshape(-1, O))            # policy.py:60-72
        mean = policy.policy_mean(states.reshape(-1, O))
        values = critic.get_value(states).squeeze(-1)
        next_values = critic.get_value(next_states).squeeze(-1)
        det = policy.get_deterministic_action(states.reshape(-1, O))
        action, logp, values, next_values = R(action), R(logp), R(values), R(next_values)       # inputs of the next stages
        adv, ret = gae_fn(rewards, term, values, next_values, hp["gamma"], hp["gae_lambda"])   # ppo.py:109-118
    if round_inputs:
        out["advantages_exact"], out["returns_exact"] = adv, ret                               # GAE outputs before any rounding
    adv, ret = R(adv), R(ret)
    out.update(states=states, next_states=next_states, rewards=rewards, terminations=term, actions=action.reshape(T, N, A),
               scaled_actions=scaled.reshape(T, N, A), log_probs=logp.reshape(T, N), mean=mean.reshape(T, N, A),
               deterministic_actions=det.reshape(T, N, A), values=values, next_values=next_values, advantages=adv, returns=ret)
    # the policy moves away from the behaviour policy (ratio != 1, some samples clipped)
    jitter(policy, gen, 0.02)
    round_module(policy, R)
    out["pparams1"] = flat_mlp(plin, [policy.policy_logstd])
    bs, ba = states.reshape(-1, O), action.reshape(-1, A)
    badv, bret, blp = adv.reshape(-1), ret.reshape(-1), logp.reshape(-1)
    perm = torch.randperm(T * N, generator=gen)
    for step in range(2):                                                                  # two consecutive minibatch updates
        idx = perm[step * MB:(step + 1) * MB]
        with torch.no_grad():
            nlp, ent = policy.get_logprob_entropy(bs[idx], ba[idx])                          # policy.py:75-81
        pg, el, kl, cf, pgn = policy_loss_fn(bs[idx], ba[idx], blp[idx], badv[idx])         # ppo.py:121-150
        cl, cgn = critic_loss_fn(bs[idx], bret[idx])                                        # ppo.py:153-166
        s = "_%d" % step
        out.update({"idx" + s: idx.numpy().astype(np.int32), "new_log_prob" + s: nlp, "entropy" + s: ent,
                    "pg_loss" + s: pg.detach(), "entropy_loss" + s: el.detach(), "approx_kl" + s: kl, "clip_fraction" + s: cf,
                    "policy_grad_norm" + s: pgn, "critic_loss" + s: cl.detach(), "critic_grad_norm" + s: cgn,
                    "pgrads_clipped" + s: flat_grads(plin, [policy.policy_logstd]), "cgrads_clipped" + s: flat_grads(clin),
                    "pparams_after" + s: flat_mlp(plin, [policy.policy_logstd]), "cparams_after" + s: flat_mlp(clin)})
    save("reference_ppo_%s.npz" % tag, out)


# ------------------------------------------------------------------------------------- PPO, Categorical policy
def make_ppo_discrete(dtype, tag, round_inputs=False):
    """DiscreteFlatValuesPolicy (ppo/pytorch/policy.py:96-135) + the same policy_loss_fn / critic_loss_fn closures."""
    R = rounder(dtype, round_inputs)
    sys.path.insert(0, REF)
    pol_mod = load_by_path("rl_x/algorithms/ppo/pytorch/policy.py", "ref_ppo_policy")
    cri_mod = load_by_path("rl_x/algorithms/ppo/pytorch/critic.py", "ref_ppo_critic")
    O, NA, H, B, MB = 4, 3, 64, 192, 64
    torch.manual_seed(7)
    gen = torch.Generator().manual_seed(9)
    env = fake_env(O, 1)
    env.get_single_action_logit_size = lambda: NA
    policy = pol_mod.DiscreteFlatValuesPolicy(env, H, "cpu", np.arange(O)).to(dtype)
    critic = cri_mod.FlatValuesCritic(env, H, "cpu", np.arange(O)).to(dtype)
    jitter(policy, gen, 0.3)            # the 0.01-scaled logits head would give near-uniform probabilities
    jitter(critic, gen, 0.05)
    round_module(policy, R)
    round_module(critic, R)
    plin = [m for m in policy.policy_mean if isinstance(m, torch.nn.Linear)]
    clin = [m for m in critic.critic if isinstance(m, torch.nn.Linear)]
    hp = dict(clip_range=0.2, entropy_coef=0.01, critic_coef=0.5, max_grad_norm=0.5, learning_rate=3e-4)
    self = types.SimpleNamespace(policy=policy, critic=critic, bf16_mixed_precision_training=False, compile_mode="default", **hp)
    self.policy_optimizer = torch.optim.Adam(policy.parameters(), lr=hp["learning_rate"], fused=False)
    self.critic_optimizer = torch.optim.Adam(critic.parameters(), lr=hp["learning_rate"], fused=False)
    ns = {"torch": torch, "nn": torch.nn, "autocast": torch.amp.autocast, "self": self}
    policy_loss_fn, critic_loss_fn = train_closures("rl_x/algorithms/ppo/pytorch/ppo.py", ["policy_loss_fn", "critic_loss_fn"], ns)
    r = lambda *s: R(torch.randn(*s, generator=gen, dtype=dtype))
    states = r(B, O)
    out = {"obs_dim": O, "nr_actions": NA, "hidden": H, "source": "reference:rl_x/algorithms/ppo/pytorch (executed)", **hp}
    out["pparams0"] = flat_mlp(plin)
    out["cparams0"] = flat_mlp(clin)
    with torch.no_grad():
        action, processed, logp = policy.get_action_logprob(states)                          # policy.py:118-124
        logits = policy.policy_mean(states)
        det = policy.get_deterministic_action(states)
        logp = R(logp)
    out.update(states=states, actions=action.to(torch.int64), logits=logits, log_probs=logp, deterministic_actions=det.to(torch.int64))
    jitter(policy, gen, 0.03)
    round_module(policy, R)
    out["pparams1"] = flat_mlp(plin)
    adv, ret = R(2 * r(B) + 0.3), r(B)
    out.update(advantages=adv, returns=ret)
    perm = torch.randperm(B, generator=gen)
    for step in range(2):
        idx = perm[step * MB:(step + 1) * MB]
        with torch.no_grad():
            nlp, ent = policy.get_logprob_entropy(states[idx], action[idx])                  # policy.py:126-130
        pg, el, kl, cf, pgn = policy_loss_fn(states[idx], action[idx], logp[idx], adv[idx])  # ppo.py:121-150
        cl, cgn = critic_loss_fn(states[idx], ret[idx])
        s = "_%d" % step
        out.update({"idx" + s: idx.numpy().astype(np.int32), "new_log_prob" + s: nlp, "entropy" + s: ent,
                    "pg_loss" + s: pg.detach(), "entropy_loss" + s: el.detach(), "approx_kl" + s: kl, "clip_fraction" + s: cf,
                    "policy_grad_norm" + s: pgn, "critic_loss" + s: cl.detach(), "critic_grad_norm" + s: cgn,
                    "pgrads_clipped" + s: flat_grads(plin), "cgrads_clipped" + s: flat_grads(clin),
                    "pparams_after" + s: flat_mlp(plin), "cparams_after" + s: flat_mlp(clin)})
    save("reference_ppo_discrete_%s.npz" % tag, out)


# ------------------------------------------------------------------------------------------------ SAC
def make_sac(dtype, tag, round_inputs=False):
    import torch.nn.functional as F
    R = rounder(dtype, round_inputs)
    sys.path.insert(0, REF)
    pol_mod = load_by_path("rl_x/algorithms/sac/pytorch/policy.py", "ref_sac_policy")
    q_mod = load_by_path("rl_x/algorithms/sac/pytorch/q_network.py", "ref_sac_q")
    ent_mod = load_by_path("rl_x/algorithms/sac/pytorch/entropy_coefficient.py", "ref_sac_alpha")
    O, A, H, B = 13, 4, 64, 96
    torch.manual_seed(4)
    gen = torch.Generator().manual_seed(6)
    env = fake_env(O, A)
    hp = dict(gamma=0.99, tau=0.005, log_std_min=-20.0, log_std_max=2.0, learning_rate=3e-4)
    policy = pol_mod.Policy(env, hp["log_std_min"], hp["log_std_max"], H, "cpu", np.arange(O)).to(dtype)
    qs = [q_mod.QNetwork(env, H, "cpu", np.arange(O)).to(dtype) for _ in range(4)]         # q1, q2, q1_target, q2_target
    for q in qs[2:]:
        jitter(q, gen, 0.01)                                                              # targets differ from the online nets
    cfg = types.SimpleNamespace(algorithm=types.SimpleNamespace(target_entropy="auto"))
    alpha = ent_mod.EntropyCoefficient(cfg, env, "cpu").to(dtype)
    with torch.no_grad():
        alpha.log_alpha.fill_(-0.3)
        policy.log_std.weight.mul_(0.1)           # std ~ 1: fp32 log(1 - tanh^2 + 1e-6) stays well conditioned
        policy.mean.weight.mul_(0.5)
    for mod in [policy] + qs:
        round_module(mod, R)
    critic = types.SimpleNamespace(q1=qs[0], q2=qs[1], q1_target=qs[2], q2_target=qs[3])
    self = types.SimpleNamespace(policy=policy, critic=critic, entropy_coefficient=alpha, gamma=hp["gamma"],
                                 bf16_mixed_precision_training=False, compile_mode="default")
    lr = hp["learning_rate"]
    self.policy_optimizer = torch.optim.Adam(policy.parameters(), lr=lr, fused=False)
    self.q_optimizer = torch.optim.Adam(list(qs[0].parameters()) + list(qs[1].parameters()), lr=lr, fused=False)
    self.entropy_optimizer = torch.optim.Adam(alpha.parameters(), lr=lr, fused=False)
    ns = {"torch": torch, "nn": torch.nn, "F": F, "autocast": torch.amp.autocast, "self": self}
    pe_loss_fn, q_loss_fn = train_closures("rl_x/algorithms/sac/pytorch/sac.py", ["policy_and_entropy_loss_fn", "critic_loss_fn"], ns)

    plin = [m for m in policy.torso if isinstance(m, torch.nn.Linear)]

    def policy_flat(grads=False):
        # head = [mean | log_std] side by side: one [H, 2A] matrix, bias [2A]  (the JAX flavour's Dense(2A) + split)
        f = (lambda p: p.grad) if grads else (lambda p: p.detach())
        parts = []
        for lin in plin:
            parts += [f(lin.weight).T.contiguous().reshape(-1), f(lin.bias).reshape(-1)]
        parts += [torch.cat([f(policy.mean.weight).T, f(policy.log_std.weight).T], dim=1).contiguous().reshape(-1),
                  torch.cat([f(policy.mean.bias), f(policy.log_std.bias)])]
        return torch.cat(parts).numpy().copy()

    qlin = [[m for m in q.critic if isinstance(m, torch.nn.Linear)] for q in qs]
    r = lambda *s: R(torch.randn(*s, generator=gen, dtype=dtype))
    s, s2 = r(B, O), r(B, O)
    a = R(torch.tanh(r(B, A)))
    rew = r(B)
    done = (torch.rand(B, generator=gen) < 0.2).to(dtype)
    out = {"obs_dim": O, "act_dim": A, "hidden": H, "source": "reference:rl_x/algorithms/sac/pytorch (executed)",
           "target_entropy": alpha.target_entropy, "log_alpha": -0.3, **hp}
    out.update(states=s, next_states=s2, actions=a, rewards=rew, terminations=done, pparams=policy_flat(),
               qparams=np.concatenate([flat_mlp(qlin[0]), flat_mlp(qlin[1])]),
               qtarget=np.concatenate([flat_mlp(qlin[2]), flat_mlp(qlin[3])]))
    import torch.distributions.normal as tdn
    raw_normal = torch.distributions.utils._standard_normal
    std_normal = lambda shape, dtype, device: R(raw_normal(shape, dtype, device))   # the draw Normal.rsample consumes (rounded in "f64r")
    tdn._standard_normal = std_normal
    # Policy.get_action (policy.py:45-64) draws its noise through Normal.rsample -> _standard_normal(shape): replay the
    # generator state to record the very noise each call consumed
    st = torch.get_rng_state()
    out["noise_next"] = std_normal((B, A), dtype, torch.device("cpu"))
    torch.set_rng_state(st)
    with torch.no_grad():
        a_t, a_sc, lp = policy.get_action(s2)
        out.update(q1=qs[0](s, a).reshape(-1), q2=qs[1](s, a).reshape(-1))                   # q_network.py:38-41
    out.update(next_action=a_t, next_scaled_action=a_sc, next_log_prob=lp.reshape(-1), deterministic_action=policy.get_deterministic_action(s2).detach())
    torch.set_rng_state(st)
    q_loss, q_gn = q_loss_fn(s, s2, a, rew, done)                                          # sac.py:129-166 (consumes the same noise)
    out.update(q_loss=q_loss.detach(), critic_grad_norm=q_gn, gcritic=np.concatenate([flat_grads(qlin[0]), flat_grads(qlin[1])]),
               qparams_after=np.concatenate([flat_mlp(qlin[0]), flat_mlp(qlin[1])]))
    # policy / alpha losses against the critics of the fixture (restore them: the JAX flavour differentiates all three
    # losses at the SAME parameters)
    qp = out["qparams"]
    with torch.no_grad():
        off = 0
        for lins in qlin[:2]:
            for lin in lins:
                n = lin.weight.numel()
                lin.weight.copy_(torch.from_numpy(qp[off:off + n]).reshape(lin.in_features, lin.out_features).T); off += n
                lin.bias.copy_(torch.from_numpy(qp[off:off + lin.out_features])); off += lin.out_features
    st = torch.get_rng_state()
    out["noise_cur"] = std_normal((B, A), dtype, torch.device("cpu"))
    torch.set_rng_state(st)
    p_loss, e_loss, min_q, ent_mean, alpha_d, p_gn, e_gn = pe_loss_fn(s)                   # sac.py:90-126
    out.update(policy_loss=p_loss.detach(), entropy_loss=e_loss.detach(), min_q_mean=min_q.detach().mean(), entropy=ent_mean,
               alpha=alpha_d, policy_grad_norm=p_gn, g_log_alpha=alpha.log_alpha.grad.reshape(()), gpolicy=policy_flat(True),
               pparams_after=policy_flat(), log_alpha_after=alpha.log_alpha.detach().reshape(()))
    tdn._standard_normal = raw_normal
    save("reference_sac_%s.npz" % tag, out)


# --------------------------------------------------------------------------------------- SAC replay ring
def make_replay():
    """rl_x/algorithms/sac/flax/replay_buffer.py is numpy only: the JAX flavour's own class runs here unchanged."""
    rb_mod = load_by_path("rl_x/algorithms/sac/flax/replay_buffer.py", "ref_sac_replay")
    O, A, NE, cap_rows, steps, B = 5, 2, 6, 7, 11, 64                          # 11 adds into 7 rows: the ring wraps
    data = np.random.default_rng(21)
    rb = rb_mod.ReplayBuffer(cap_rows * NE + 3, NE, (O,), (A,), np.random.default_rng(8))   # capacity // nr_envs rounds down
    adds = dict(states=data.standard_normal((steps, NE, O)).astype(np.float32), next_states=data.standard_normal((steps, NE, O)).astype(np.float32),
                actions=data.standard_normal((steps, NE, A)).astype(np.float32), rewards=data.standard_normal((steps, NE)).astype(np.float32),
                terminations=(data.random((steps, NE)) < 0.3).astype(np.float32))
    out = {"source": "reference:rl_x/algorithms/sac/flax/replay_buffer.py (executed)", "capacity": cap_rows * NE + 3, "nr_envs": NE,
           "sampler_seed": 8, "batch": B}
    out.update({"add_" + k: v for k, v in adds.items()})
    for t in range(steps):
        rb.add(*(adds[k][t] for k in ("states", "next_states", "actions", "rewards", "terminations")))
        if t in (3, steps - 1):                                                # once before the ring is full, once after it wrapped
            smp = rb.sample(B)
            out.update({"sample%d_%s" % (t, k): v for k, v in zip(("states", "next_states", "actions", "rewards", "terminations"), smp)})
    out.update(final_pos=rb.pos, final_size=rb.size, ring_states=rb.states)
    save("reference_sac_replay.npz", out)


def make_obs_norm():
    """rl_x/algorithms/fastsac/pytorch/observation_normalizer.py needs torch alone: the module itself runs here (its
    `torch.compile` wrapper get_observation_normalizer is not called -- it does not change the arithmetic)."""
    mod = load_by_path("rl_x/algorithms/fastsac/pytorch/observation_normalizer.py", "ref_obs_norm")
    O = 7
    data = np.random.default_rng(33)
    batches = [(data.standard_normal((n, O)) * data.uniform(0.1, 30.0, O) + data.uniform(-50.0, 50.0, O)).astype(np.float32)
               for n in (256, 1, 97, 1024)]                                  # a one-row batch (variance 0) among them
    batches.append(np.concatenate([batches[0][:5, :3], np.full((5, 4), 2.5, np.float32)], axis=1))   # constant columns
    out = {"source": "reference:rl_x/algorithms/fastsac/pytorch/observation_normalizer.py (executed)", "epsilon": 1e-8}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dtype)
        nrm = mod.ObservationNormalizer(O, "cpu", True)
        nrm.train()
        for i, b in enumerate(batches):
            y = nrm.normalize(torch.as_tensor(b, dtype=dtype), update=True)
            out.update({"%s_out%d" % (tag, i): y.clone(), "%s_mean%d" % (tag, i): nrm.running_mean.clone(),
                        "%s_var%d" % (tag, i): nrm.running_var.clone(), "%s_std%d" % (tag, i): nrm.running_std_dev.clone(),
                        "%s_count%d" % (tag, i): nrm.count.clone()})
        probe = torch.as_tensor(batches[2], dtype=dtype)
        out["%s_frozen" % tag] = nrm.normalize(probe, update=False).clone()    # acting / evaluation: no update (fastsac.py:253)
        nrm.eval()
        out["%s_eval" % tag] = nrm.normalize(probe, update=True).clone()       # eval mode never updates (:29)
        out["%s_count_final" % tag] = nrm.count.clone()
        torch.set_default_dtype(torch.float32)
    out.update({"batch%d" % i: b for i, b in enumerate(batches)})
    save("reference_obs_norm.npz", out)


def make_c51():
    """FastSAC's distributional critic step: the closure `critic_and_entropy_loss_fn` of `FastSAC.train`
    (rl_x/algorithms/fastsac/pytorch/fastsac.py:144-241) compiled from the reference file and run against a stand-in `self`
    whose four Q networks are TABLES of logits (a module holding one [B, nr_atoms] parameter and returning it whatever the
    input): the categorical projection, the double-Q selection and the cross-entropy are then exercised exactly as written, and
    the parameter gradients ARE d q_loss / d logits.  Policy and entropy coefficient are stand-ins with fixed outputs."""
    import torch.nn as nn
    import torch.nn.functional as F

    class Table(nn.Module):
        def __init__(self, logits):
            super().__init__()
            self.logits = nn.Parameter(logits.clone())

        def forward(self, states, actions):
            return self.logits

    class Alpha(nn.Module):
        def __init__(self, log_alpha, target_entropy):
            super().__init__()
            self.log_alpha = nn.Parameter(torch.tensor([log_alpha], dtype=torch.get_default_dtype()))
            self.target_entropy = target_entropy

        def forward(self):
            return self.log_alpha.exp()

        def loss(self, entropy):
            return self.log_alpha.exp() * (entropy - self.target_entropy)

    out = {"source": "reference:rl_x/algorithms/fastsac/pytorch/fastsac.py critic_and_entropy_loss_fn (executed)"}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        torch.set_default_dtype(dtype)
        for case, (B, NA, vmin, vmax, clipped, nstep) in enumerate(((48, 101, -20.0, 20.0, False, 1), (32, 51, -5.0, 7.0, True, 3),
                                                                    (16, 11, -1.0, 1.0, True, 1))):
            g = torch.Generator().manual_seed(40 + case)
            rnd = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64).to(dtype)
            logits = [rnd(B, NA) * 1.5 for _ in range(4)]          # q1, q2, q1_target, q2_target
            rewards = rnd(B) * (0.4 * (vmax - vmin))               # many targets clamp at v_min / v_max
            dones = (torch.rand(B, generator=g) < 0.3).to(dtype)
            truncs = ((torch.rand(B, generator=g) < 0.5).to(dtype)) * dones
            nsteps = torch.randint(1, nstep + 1, (B,), generator=g).to(dtype)
            next_logp = rnd(B) - 2.0
            if case == 2:   # atoms that land EXACTLY on a support point (the l == u branch): gamma^n * z + r on the grid
                rewards = torch.zeros(B, dtype=dtype)
                next_logp = torch.zeros(B, dtype=dtype)
                dones[: B // 2] = 1.0
                truncs[: B // 2] = 0.0                             # discount 0: every atom maps to r = 0, the middle support point
            me = types.SimpleNamespace()
            me.v_min, me.v_max, me.nr_atoms, me.gamma = vmin, vmax, NA, 0.99 if case < 2 else 1.0
            me.clipped_double_q_learning, me.bf16_mixed_precision_training, me.max_grad_norm = clipped, False, -1.0
            me.device = torch.device("cpu")
            me.q_support = torch.linspace(vmin, vmax, NA)
            me.critic = types.SimpleNamespace(q1=Table(logits[0]), q2=Table(logits[1]), q1_target=Table(logits[2]), q2_target=Table(logits[3]))
            me.policy = types.SimpleNamespace(get_action_and_log_prob=lambda s, lp=next_logp: (torch.zeros(s.shape[0], 1), lp))
            me.entropy_coefficient = Alpha(-0.7, -3.0)
            me.q_optimizer = torch.optim.SGD(list(me.critic.q1.parameters()) + list(me.critic.q2.parameters()), lr=0.0)
            me.entropy_optimizer = torch.optim.SGD([me.entropy_coefficient.log_alpha], lr=0.0)
            ns = {"torch": torch, "F": F, "self": me, "autocast": torch.autocast}
            fn = train_closures("rl_x/algorithms/fastsac/pytorch/fastsac.py", ["critic_and_entropy_loss_fn"], ns)[0]
            st = torch.zeros(B, 3)
            q_loss, ent_loss, q_min, q_max, ent_mean, gnorm, _ = fn(st, st, torch.zeros(B, 1), rewards, dones, truncs, nsteps)
            k = "%s_c%d_" % (tag, case)
            out.update({k + "q1": logits[0], k + "q2": logits[1], k + "q1_target": logits[2], k + "q2_target": logits[3],
                        k + "rewards": rewards, k + "dones": dones, k + "truncations": truncs, k + "n_steps": nsteps,
                        k + "next_log_probs": next_logp, k + "alpha": me.entropy_coefficient().detach(), k + "gamma": me.gamma,
                        k + "v_min": vmin, k + "v_max": vmax, k + "clipped": int(clipped),
                        k + "q_loss": q_loss.detach(), k + "q_min": q_min.detach(), k + "q_max": q_max.detach(),
                        k + "d_q1": me.critic.q1.logits.grad.clone(), k + "d_q2": me.critic.q2.logits.grad.clone(),
                        k + "grad_norm": torch.as_tensor(gnorm).detach()})
        torch.set_default_dtype(torch.float32)
    out["n_cases"] = 3
    save("reference_c51.npz", out)



# ------------------------------------------------------------------------------------------------ FastSAC
def fastsac_params(seed, O, A, NA):
    """The seeded numpy parameters of oracle/fastsac.py (make_params): a test rebuilds exactly these arrays."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.fastsac import make_params
    return make_params(seed, O, A, NA)


def _load_lnmlp(seq, heads, flat, in_dim, hidden, dtype):
    """flat layout -> the reference's nn.Sequential(Linear, LayerNorm, SiLU, ...) + head Linear(s) (column blocks of the head)."""
    lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
    lns = [m for m in seq if isinstance(m, torch.nn.LayerNorm)]
    off, d = 0, in_dim
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    with torch.no_grad():
        for li, h in enumerate(hidden):
            lins[li].weight.copy_(t(flat[off:off + d * h].reshape(d, h).T)); off += d * h
            lins[li].bias.copy_(t(flat[off:off + h])); off += h
            lns[li].weight.copy_(t(flat[off:off + h])); off += h
            lns[li].bias.copy_(t(flat[off:off + h])); off += h
            d = h
        width = sum(hd.out_features for hd in heads)
        W = flat[off:off + d * width].reshape(d, width); off += d * width
        b = flat[off:off + width]; off += width
        c = 0
        for hd in heads:
            hd.weight.copy_(t(W[:, c:c + hd.out_features].T))
            hd.bias.copy_(t(b[c:c + hd.out_features]))
            c += hd.out_features
    assert off == flat.size, (off, flat.size)


def _flat_lnmlp(seq, heads, grads=False):
    f = (lambda p: p.grad) if grads else (lambda p: p.detach())
    lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
    lns = [m for m in seq if isinstance(m, torch.nn.LayerNorm)]
    parts = []
    for lin, ln in zip(lins, lns):
        parts += [f(lin.weight).T.contiguous().reshape(-1), f(lin.bias).reshape(-1), f(ln.weight).reshape(-1), f(ln.bias).reshape(-1)]
    parts += [torch.cat([f(h.weight).T for h in heads], dim=1).contiguous().reshape(-1), torch.cat([f(h.bias) for h in heads])]
    return torch.cat(parts).to(torch.float64).numpy().copy()


def _sampled(name, full, rng_seed, n=1500):
    """A fixture stays small: the L2 norm of the vector, and n entries at seeded positions."""
    idx = np.random.default_rng(rng_seed).choice(full.size, size=min(n, full.size), replace=False)
    idx.sort()
    return {name + "_norm": np.linalg.norm(full), name + "_idx": idx.astype(np.int64), name + "_val": full[idx]}


def make_fastsac():
    """One critic step, the Polyak update and one policy step of FastSAC as the reference computes them: the modules
    `Policy` (fastsac/pytorch/policy.py:24-108), `QNetwork` (q_network.py:20-43), `EntropyCoefficient`
    (entropy_coefficient.py:16-30) and the closures `critic_and_entropy_loss_fn` / `policy_loss_fn` of `FastSAC.train`
    (fastsac.py:105-241), executed in float64 on float32-representable inputs, with AdamW optimisers built as the reference
    builds them (fastsac.py:88-91; fused=False on the CPU).  Parameters come from `fastsac_params` (numpy, seeded): a test
    regenerates them; the file holds the batch, the noise the two `rsample` calls consumed, scalars, and norms + seeded samples of
    the gradients and of the updated parameters."""
    import torch.nn.functional as F
    import torch.distributions.normal as tdn
    sys.path.insert(0, REF)
    pol_mod = load_by_path("rl_x/algorithms/fastsac/pytorch/policy.py", "ref_fsac_policy")
    q_mod = load_by_path("rl_x/algorithms/fastsac/pytorch/q_network.py", "ref_fsac_q")
    ent_mod = load_by_path("rl_x/algorithms/fastsac/pytorch/entropy_coefficient.py", "ref_fsac_alpha")
    dtype = torch.float64
    torch.set_default_dtype(dtype)
    out = {"source": "reference:rl_x/algorithms/fastsac/pytorch (executed)", "n_cases": 3}
    # case 2: gradient clipping on (max_grad_norm = 0.05, below both gradient norms: torch.nn.utils.clip_grad_norm_, fastsac.py:129-130, :218-219)
    for case, (O, A, NA, B, clipped, seed, mgn) in enumerate(((9, 3, 21, 48, False, 11, -1.0), (7, 2, 101, 32, True, 12, -1.0),
                                                              (9, 3, 21, 48, True, 13, 0.05))):
        # the noise of the two rsample calls comes from torch's GLOBAL generator (torch.distributions.utils._standard_normal has no
        # generator argument): seed it per case, so that the file is a function of this script alone and not of what ran before it
        torch.manual_seed(80 + case)
        sp = types.SimpleNamespace
        low, high = np.linspace(-1.0, -0.5, A), np.linspace(1.0, 2.0, A)
        center, scale = 0.5 * (low + high) * 0.5, np.linspace(1.0, 0.8, A)
        env = sp(single_action_space=sp(low=low, high=high, center=center, scale=scale, shape=(A,)), single_observation_space=sp(shape=(O,)))
        hp = dict(gamma=0.97, tau=0.125, v_min=-20.0, v_max=20.0, log_std_min=-5.0, log_std_max=0.0, learning_rate=3e-4, weight_decay=0.001,
                  adam_beta1=0.9, adam_beta2=0.95, target_entropy=0.0, log_alpha=float(np.log(0.2)))
        policy = pol_mod.Policy(env, hp["log_std_min"], hp["log_std_max"], "cpu", np.arange(O)).to(dtype)
        qs = [q_mod.QNetwork(env, NA, "cpu", np.arange(O)).to(dtype) for _ in range(4)]
        pflat, qflat = fastsac_params(seed, O, A, NA)
        _load_lnmlp(policy.torso, [policy.mean, policy.log_std], pflat, O, (512, 256, 128), dtype)
        for q, fl in zip(qs, qflat):
            _load_lnmlp(q.critic[:-1], [q.critic[-1]], fl, O + A, (768, 384, 192), dtype)
        cfg = sp(algorithm=sp(target_entropy=hp["target_entropy"], alpha_init=float(np.exp(hp["log_alpha"]))))
        alpha = ent_mod.EntropyCoefficient(cfg, env, "cpu").to(dtype)
        with torch.no_grad():
            alpha.log_alpha.fill_(float(np.float32(hp["log_alpha"])))
        critic = sp(q1=qs[0], q2=qs[1], q1_target=qs[2], q2_target=qs[3])
        me = sp(policy=policy, critic=critic, entropy_coefficient=alpha, gamma=hp["gamma"], v_min=hp["v_min"], v_max=hp["v_max"],
                nr_atoms=NA, clipped_double_q_learning=clipped, bf16_mixed_precision_training=False, max_grad_norm=mgn,
                device=torch.device("cpu"), q_support=torch.linspace(hp["v_min"], hp["v_max"], NA))
        kw = dict(lr=hp["learning_rate"], weight_decay=hp["weight_decay"], betas=(hp["adam_beta1"], hp["adam_beta2"]), fused=False)
        me.policy_optimizer = torch.optim.AdamW(policy.parameters(), **kw)
        me.q_optimizer = torch.optim.AdamW(list(qs[0].parameters()) + list(qs[1].parameters()), **kw)
        me.entropy_optimizer = torch.optim.AdamW([alpha.log_alpha], **kw)
        ns = {"torch": torch, "F": F, "self": me, "autocast": torch.autocast}
        policy_loss_fn, critic_fn = train_closures("rl_x/algorithms/fastsac/pytorch/fastsac.py",
                                                   ["policy_loss_fn", "critic_and_entropy_loss_fn"], ns)
        g = torch.Generator().manual_seed(70 + case)
        r32 = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float64).to(torch.float32).to(dtype)
        s, s2 = r32(B, O), r32(B, O)
        a = (torch.tanh(r32(B, A)) * policy.action_scale).to(torch.float32).to(dtype)
        rew = r32(B) * 3.0
        done = (torch.rand(B, generator=g) < 0.3).to(dtype)
        trunc = (torch.rand(B, generator=g) < 0.5).to(dtype) * done
        nst = torch.randint(1, 4, (B,), generator=g).to(dtype)
        raw_normal = torch.distributions.utils._standard_normal
        noise = []

        def std_normal(shape, dtype, device):
            e = raw_normal(shape, dtype, device).to(torch.float32).to(dtype)          # float32-representable noise
            noise.append(e.clone())
            return e
        tdn._standard_normal = std_normal
        k = "c%d_" % case
        out.update({k + "obs_dim": O, k + "act_dim": A, k + "nr_atoms": NA, k + "batch": B, k + "clipped": int(clipped), k + "param_seed": seed,
                    k + "action_scale": policy.action_scale.clone(), k + "states": s, k + "next_states": s2, k + "actions": a, k + "rewards": rew,
                    k + "dones": done, k + "truncations": trunc, k + "n_steps": nst})
        out.update({k + n: v for n, v in hp.items()})
        out[k + "max_grad_norm"] = mgn
        # --- acting outputs of the policy module on the fixture's parameters (policy.py:74-108)
        with torch.no_grad():
            mean, log_std = policy(s)
            out.update({k + "mean": mean, k + "log_std": log_std, k + "deterministic_action": policy.get_action(s, deterministic=True),
                        k + "q1_logits": qs[0](s, a), k + "q2_logits": qs[1](s, a)})
        # --- critic + entropy step (fastsac.py:144-241)
        q_loss, ent_loss, q_min, q_max, ent_mean, c_gn, e_gn = critic_fn(s, s2, a, rew, done, trunc, nst)
        out.update({k + "noise_next": noise[-1], k + "q_loss": q_loss.detach(), k + "entropy_loss": ent_loss.detach(), k + "q_min": q_min.detach(),
                    k + "q_max": q_max.detach(), k + "entropy": ent_mean.detach(), k + "critic_grad_norm": torch.as_tensor(c_gn).detach(),
                    k + "entropy_grad_norm": torch.as_tensor(e_gn).detach(), k + "log_alpha_after": alpha.log_alpha.detach().reshape(())})
        gq = np.concatenate([_flat_lnmlp(q.critic[:-1], [q.critic[-1]], grads=True) for q in qs[:2]])
        qa = np.concatenate([_flat_lnmlp(q.critic[:-1], [q.critic[-1]]) for q in qs[:2]])
        out.update(_sampled(k + "gcritic", gq, 100 + case))
        out.update(_sampled(k + "qparams_after", qa, 200 + case))
        # --- Polyak update of the targets (fastsac.py:323-327), as the train loop applies it after every critic step
        with torch.no_grad():
            for qo, qt in ((qs[0], qs[2]), (qs[1], qs[3])):
                for param, target_param in zip(qo.parameters(), qt.parameters()):
                    target_param.data.mul_(1.0 - hp["tau"]).add_(param.data, alpha=hp["tau"])
        ta = np.concatenate([_flat_lnmlp(q.critic[:-1], [q.critic[-1]]) for q in qs[2:]])
        out.update(_sampled(k + "qtarget_after", ta, 300 + case))
        # --- policy step on the updated critics (fastsac.py:106-141, :329)
        p_loss, alpha_d, p_gn = policy_loss_fn(s)
        out.update({k + "noise_cur": noise[-1], k + "policy_loss": p_loss.detach(), k + "alpha_at_policy_step": alpha_d.detach().reshape(()),
                    k + "policy_grad_norm": torch.as_tensor(p_gn).detach()})
        gp = _flat_lnmlp(policy.torso, [policy.mean, policy.log_std], grads=True)
        pa = _flat_lnmlp(policy.torso, [policy.mean, policy.log_std])
        out.update(_sampled(k + "gpolicy", gp, 400 + case))
        out.update(_sampled(k + "pparams_after", pa, 500 + case))
        tdn._standard_normal = raw_normal
    torch.set_default_dtype(torch.float32)
    # --- n-step replay ring (replay_buffer.py:4-96), the reference's own class; torch.randint recorded
    rb_mod = load_by_path("rl_x/algorithms/fastsac/pytorch/replay_buffer.py", "ref_fsac_replay")
    O, A, NE, cap, B = 4, 2, 5, 6, 40
    g = torch.Generator().manual_seed(90)
    out["ring_gamma"] = 0.97
    raw_randint, drawn = torch.randint, []

    def rec_randint(*a, **k):
        k.pop("device", None)
        v = raw_randint(*a, generator=g, **k)
        drawn.append(v.clone())
        return v
    for tag, n_steps, steps in (("n1", 1, 4), ("n3_partial", 3, 5), ("n3_full", 3, 9)):
        rb = rb_mod.ReplayBuffer(cap, NE, (O,), (A,), n_steps, 0.97, "cpu")
        for t_ in range(steps):
            dn = (torch.rand(NE, generator=g) < 0.35).float()
            rb.add(torch.randn(NE, O, generator=g), torch.randn(NE, O, generator=g), torch.randn(NE, A, generator=g), torch.randn(NE, generator=g),
                   dn, (torch.rand(NE, generator=g) < 0.5).float() * dn)
        torch.randint = rec_randint
        drawn.clear()
        smp = rb.sample(B)
        torch.randint = raw_randint
        out.update({tag + "_n_steps": n_steps, tag + "_pos": rb.pos, tag + "_size": rb.size, tag + "_idx_t": drawn[0], tag + "_idx_e": drawn[1]})
        out.update({tag + "_ring_" + k: getattr(rb, k).clone() for k in ("states", "next_states", "actions", "rewards", "dones", "truncations")})
        out.update({tag + "_" + k: v for k, v in zip(("states", "next_states", "actions", "rewards", "dones", "truncations", "effective_n_steps"), smp)})
    out.update({"ring_" + k: out["n1_ring_" + k] for k in ("states", "next_states", "actions", "rewards", "dones", "truncations")})
    save("reference_fastsac.npz", out)

def save(name, out):
    arrs = {}
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        arrs[k] = np.asarray(v)
    np.savez_compressed(os.path.join(os.environ.get("RLX_GOLDEN_OUT", HERE), name), **arrs)
    print("wrote", name, "(%d arrays)" % len(arrs))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference checkout is needed to regenerate these fixtures (%s)" % REF)
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        make_ppo(dtype, tag)
        make_ppo_discrete(dtype, tag)
        make_sac(dtype, tag)
    # float64 arithmetic on float32-representable inputs: what the fp32 kernels are held to 1e-5 against
    make_ppo(torch.float64, "f64r", round_inputs=True)
    make_ppo_discrete(torch.float64, "f64r", round_inputs=True)
    make_sac(torch.float64, "f64r", round_inputs=True)
    make_replay()
    make_obs_norm()
    make_c51()
    make_fastsac()
