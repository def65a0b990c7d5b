"""GPU: the fused acting step (one launch: policy+critic forward, sample, log-prob, env
transition) against the unfused kernels and the oracle."""
import numpy as np
import pytest
import torch

from oracle import env as oenv, nets, ppo as oppo, prng
from rlx_amd.hip import mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("arch,N,O,A", [("B", 256, 17, 6), ("A", 100, 17, 6), ("B", 4096, 17, 6), ("B", 70, 4, 2)])
@pytest.mark.parametrize("scheme", [1, 0])
@pytest.mark.parametrize("images", [False, True], ids=["fp32-layers", "fp16-pipe-layers"])
def test_fused_step_matches_oracle(ctx, dev, arch, N, O, A, scheme, images):
    rng = np.random.default_rng(N + O)
    ps, cs = nets.make_spec(arch, O, A, True), nets.make_spec(arch, O, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.05 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    pd = mlp_desc(O, ps.hidden, A, ps.act, ps.ln_first, True)
    cd = mlp_desc(O, cs.hidden, 1, cs.act, cs.ln_first, False)
    assert ctx.rollout_step_supported(pd, cd)
    seed, horizon, p_term, noise, off = 9, 6, 0.1, 0.1, 128
    o = oenv.RandomObsEnvOracle(seed, N, O, A, horizon, p_term, noise, off)
    obs_e = o.reset()
    obs_a, obs_b = _t(obs_e, dev), torch.empty(N, O, device=dev)
    ep_step = _t(o.ep_step, dev)
    ep_ret, last_ret, last_len = (torch.zeros(N, device=dev) for _ in range(3))
    stats = torch.zeros(4, device=dev)
    action, value, logp = torch.empty(N, A, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    fin, rew, term = torch.empty(N, O, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    key = prng.prng_key(5)
    P, C = _t(pp, dev), _t(cp, dev)
    ndone = 0
    if images:      # hidden layers on the fp16 matrix pipe from the weight images laid out once per rollout
        ctx.rollout_begin(pd, P, cd, C)
    else:
        ctx.rollout_end()
    for t in range(8):
        env = dict(seed=seed, env_id_offset=off, t=t, horizon=horizon, p_term=p_term, reward_noise=noise,
                   final_obs=fin, reward=rew, terminated=term, ep_step=ep_step, ep_ret=ep_ret, last_ret=last_ret,
                   last_len=last_len, episode_stats=stats)
        obs_in = obs_a.cpu().numpy()     # the nets are compared on IDENTICAL inputs (the env's own parity is asserted below)
        new_key = ctx.rollout_step(pd, P, cd, C, obs_a, obs_b, key, action, None, value, logp, scheme=scheme,
                                   noise_row_offset=off, n_global=N + off + 3, env=env)
        ks = prng.split(key, 2, bool(scheme))
        assert np.array_equal(new_key, ks[0])
        eps = prng.normal(ks[1], (N + off + 3, A), bool(scheme))[off:off + N]
        a_e, v_e, lp_e = oppo.get_action_and_value(ps, pp.astype(np.float64), cs, cp.astype(np.float64),
                                                   obs_in.astype(np.float64), eps.astype(np.float64))
        np.testing.assert_allclose(value.cpu().numpy(), v_e, rtol=1e-5, atol=1e-5)
        # (the N(0,1) draws are erfinv of float32 uniforms: ill-conditioned in the tails, where the two float32 evaluations
        #  of the same polynomial differ by ~1.6e-5 relative at |eps| ~ 4; the uniform bits themselves are compared exactly
        #  in test_gpu_prng.py)
        np.testing.assert_allclose(action.cpu().numpy(), a_e, rtol=3e-5, atol=1e-5)
        np.testing.assert_allclose(logp.cpu().numpy(), lp_e, rtol=3e-5, atol=2e-5)
        # env transition driven by the GPU's own action (identical inputs on both sides)
        obs_e, fin_e, r_e, term_e, trunc_e, done_e = o.step(action.cpu().numpy())
        np.testing.assert_allclose(rew.cpu().numpy(), r_e, rtol=1e-5, atol=1e-5)
        assert np.array_equal(term.cpu().numpy() > 0.5, term_e)
        np.testing.assert_allclose(fin.cpu().numpy(), fin_e, rtol=2e-5, atol=2e-6)   # Box-Muller tails: v_log_f32 / v_cos_f32 vs libm
        np.testing.assert_allclose(obs_b.cpu().numpy(), obs_e, rtol=2e-5, atol=2e-6)
        assert np.array_equal(ep_step.cpu().numpy(), o.ep_step)
        ndone += int(done_e.sum())
        key = new_key
        obs_a, obs_b = obs_b, obs_a
    assert ndone > 0 and stats[0].item() == ndone


def test_fused_rollout_equals_unfused_training(monkeypatch):
    """The whole training loop with and without the fused acting kernel: same key stream, same
    rollouts (1e-5), same learning signal."""
    import sys
    from rlx_amd.runner.runner import Runner
    res = []
    for fused in ("true", "false"):
        monkeypatch.setattr(sys, "argv", ["x", "--runner.mode=train", "--environment.nr_envs=128",
                                          "--algorithm.nr_steps=16", "--algorithm.minibatch_size=512",
                                          "--algorithm.nr_epochs=2", "--algorithm.total_timesteps=2048",
                                          f"--algorithm.fused_rollout={fused}"])
        m = Runner().run()
        res.append((m.key.copy(), m.last_metrics, m.pparams.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0])
    for k in ("loss/critic_loss", "loss/policy_gradient_loss", "v_value/explained_variance"):
        assert res[0][1][k] == pytest.approx(res[1][1][k], rel=2e-3, abs=2e-4), k
    assert np.abs(res[0][2] - res[1][2]).mean() < 1e-5


def test_one_call_rollout_equals_the_step_calls_bit_for_bit():
    """rlx_ppo_rollout_f32 (include/rlx_hip.h) queues the T launches of the acting loop itself: every rollout array, the env state
    and the key are bit-identical to T rlx_ppo_rollout_step_f32 calls (the plugin's rollout_one_call = False path)."""
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.ppo.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    res = []
    for one_call in (True, False):
        config = ConfigDict()
        config.runner = runner_cfg("train")
        config.algorithm = get_algorithm_config("ppo.hip")
        config.environment = get_environment_config("synthetic.random_obs")
        config.environment.nr_envs, config.environment.horizon = 300, 20
        config.environment.termination_probability = 0.02
        config.algorithm.nr_steps, config.algorithm.minibatch_size = 24, 1800
        config.algorithm.rollout_one_call = one_call
        env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
        model = get_algorithm_model_class("ppo.hip")(config, env, env, "/tmp/rlx_oc", None)
        assert model.rollout_one_call == one_call
        batch = model._alloc_batch()
        state, _ = env.reset()
        state = state.contiguous()
        for _ in range(2):                      # two rollouts: the env clock and the key carry over
            state = model.collect_rollout(batch, state)
        res.append([x.cpu().numpy().copy() for x in (batch.states, batch.actions, batch.values, batch.log_probs, batch.next_states,
                                                     batch.rewards, batch.terminations, state, env.ep_step, env.ep_ret,
                                                     env.last_ret, env.last_len)] + [model.key.copy(), env.episode_stats.cpu().numpy().copy(), env.t])
    assert res[0][-1] == res[1][-1] == 48
    assert res[0][6].sum() > 0                 # some episodes terminated
    for a, b in zip(res[0][:-2], res[1][:-2]):
        assert np.array_equal(a, b)
    # (the episode statistics are float atomics over the finished episodes of a step: sums equal up to their order)
    np.testing.assert_allclose(res[0][-2], res[1][-2], rtol=1e-5)


def test_advantages_value_reuse_equals_full_critic_pass():
    """compute_advantages reuses the rollout's values[t+1] for next_values[t] wherever next_states[t] == states[t+1] and
    sends only the final-observation rows (+ the last step) through the critic: same advantages as the full pass."""
    import sys as _sys
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.ppo.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config("ppo.hip")
    config.environment = get_environment_config("synthetic.random_obs")
    config.environment.nr_envs, config.environment.horizon = 512, 200
    config.environment.termination_probability = 0.01
    config.algorithm.nr_steps, config.algorithm.minibatch_size = 32, 2048
    env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    model = get_algorithm_model_class("ppo.hip")(config, env, env, "/tmp/rlx_vr", None)
    batch = model._alloc_batch()
    state, _ = env.reset()
    model.collect_rollout(batch, state.contiguous())
    differs = (batch.next_states[:-1] != batch.states[1:]).any(dim=2)
    n_diff = int(differs.sum())
    assert 0 < n_diff < 512 * 32 // 8          # some episodes ended, and the shortcut is taken
    model.compute_advantages(batch)
    adv1, ret1, nv1 = batch.advantages.clone(), batch.returns.clone(), batch.next_values.clone()
    T, N, O = 32, 512, model.obs_dim
    model.ctx.mlp_fwd(model.cdesc, model.cparams, batch.next_states.view(T * N, O), batch.next_values.view(T * N, 1))
    model.ctx.gae(batch.rewards, batch.values, batch.next_values, batch.terminations, batch.advantages, batch.returns,
                  model.gamma, model.gae_lambda)
    # (the reused values come from the acting kernel, whose hidden layers run on the fp16 pipe with split operands; the full
    #  pass below is the exact-fp32 engine: two fp32-accurate evaluations of the same critic, a few 1e-6 apart)
    torch.testing.assert_close(nv1, batch.next_values, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(adv1, batch.advantages, rtol=1e-5, atol=2e-4)   # GAE sums ~1/(1 - gamma lambda) value differences
    torch.testing.assert_close(ret1, batch.returns, rtol=1e-5, atol=2e-4)
