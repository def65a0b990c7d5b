"""GPU: error behaviour and degenerate inputs of the C ABI (include/rlx_hip.h): every entry point returns a non-zero code
and leaves a message in rlx_last_error() for arguments it cannot serve -- it never falls back to another path -- empty
inputs are no-ops, and the context stays usable after a refused call."""
import ctypes

import numpy as np
import pytest
import torch

from rlx_amd.hip import PpoHparams, mlp_desc
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu
ACT_TANH, ACT_ELU = 0, 1


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_empty_inputs_are_no_ops(ctx, dev):
    lib = ctx.lib
    d = mlp_desc(17, [64, 64], 6, ACT_TANH, False, True)
    params = torch.zeros(lib.rlx_mlp_param_count(ctypes.byref(d)), device=dev)
    x, out = torch.zeros(4, 17, device=dev), torch.full((4, 6), 7.0, device=dev)
    assert lib.rlx_mlp_fwd_f32(ctx.h, ctypes.byref(d), _p(params), _p(x), _p(out), 0, _stream()) == 0     # n = 0 rows
    a = torch.full((4,), 3.0, device=dev)
    assert lib.rlx_gae_f32(ctx.h, _p(a), _p(a), _p(a), _p(a), _p(a), _p(a), 0, 4, 0.99, 0.9, _stream()) == 0   # T = 0
    assert lib.rlx_gae_f32(ctx.h, _p(a), _p(a), _p(a), _p(a), _p(a), _p(a), 4, 0, 0.99, 0.9, _stream()) == 0   # N = 0
    bits = torch.full((4,), 5, dtype=torch.int32, device=dev)
    key = (ctypes.c_uint32 * 2)(1, 2)
    assert lib.rlx_random_bits_u32(ctx.h, key, _p(bits), 0, 1, _stream()) == 0
    assert lib.rlx_normal_f32(ctx.h, key, _p(out), 0, 1, _stream()) == 0
    torch.cuda.synchronize()
    assert bool((out == 7.0).all()) and bool((a == 3.0).all()) and bool((bits == 5).all())                 # nothing was written


@pytest.mark.parametrize("hidden,msg", [([100, 64], "hidden"), ([64, 64, 64, 64], "n_hidden"), ([64, 30], "multiples of 4")])
def test_unsupported_network_shapes_are_refused(ctx, dev, hidden, msg):
    d = mlp_desc(17, hidden, 6, ACT_TANH, False, False)
    x, out = torch.zeros(8, 17, device=dev), torch.zeros(8, 6, device=dev)
    params = torch.zeros(100000, device=dev)
    with pytest.raises(L.RlxError) as e:
        ctx.mlp_fwd(d, params, x, out)
    assert msg in str(e.value) and "rc=" in str(e.value)


def test_bad_arguments_are_refused_and_the_context_survives(ctx, dev):
    O, A, T, N = 17, 6, 4, 64
    pd = mlp_desc(O, [64, 64], A, ACT_TANH, False, True)
    cd = mlp_desc(O, [64, 64], 1, ACT_TANH, False, False)
    n_p, n_c = (ctx.lib.rlx_mlp_param_count(ctypes.byref(d)) for d in (pd, cd))
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    P, C = 0.05 * torch.randn(n_p, device=dev, generator=g), 0.05 * torch.randn(n_c, device=dev, generator=g)
    z = lambda x: torch.zeros_like(x)
    states, actions = torch.randn(T, N, O, device=dev, generator=g), torch.randn(T, N, A, device=dev, generator=g)
    logp, ret, adv = (torch.randn(T, N, device=dev, generator=g) for _ in range(3))
    hp = PpoHparams(0.2, 0.0, 0.5, 0.5, 0.9, 0.999, 1e-8)
    lr = np.full(8, 3e-4, np.float32)
    met = torch.zeros(8, 10, device=dev)
    with pytest.raises(L.RlxError, match="minibatch"):                       # 256 rows are not a multiple of 100
        ctx.ppo_update(pd, P, z(P), z(P), cd, C, z(C), z(C), states, actions, logp, ret, adv, 1, 100, L.prng_key(1), 0, lr, hp, met)
    with pytest.raises(L.RlxError, match="1-based"):
        ctx.clip_adam_step(P, z(P), z(P), z(P), 0, 3e-4, 0.5)
    with pytest.raises(L.RlxError, match="unknown option"):
        ctx.set_option("no_such_option", 1)
    perm = torch.empty(16, dtype=torch.int32, device=dev)
    with pytest.raises(L.RlxError):
        ctx.permutation(L.prng_key(1), perm, 0, 16)                          # zero epochs
    bad_critic = mlp_desc(O, [64, 64], 2, ACT_TANH, False, False)            # a critic must have one output
    idx = torch.arange(64, dtype=torch.int32, device=dev)
    with pytest.raises(L.RlxError, match="critic"):
        ctx.ppo_minibatch_fwd_bwd(pd, P, z(P), bad_critic, torch.zeros(10000, device=dev), torch.zeros(10000, device=dev),
                                  torch.zeros(8, device=dev), states, actions, logp, ret, adv, idx, hp)
    with pytest.raises(L.RlxError):                                           # host tensors are refused by the binding itself
        ctx.mlp_fwd(pd, P.cpu(), states.view(-1, O), torch.zeros(T * N, A, device=dev))
    # the context is still good: a valid update after the refused calls
    key, cnt = ctx.ppo_update(pd, P, z(P), z(P), cd, C, z(C), z(C), states, actions, logp, ret, adv, 2, 64, L.prng_key(1), 0, lr, hp, met)
    torch.cuda.synchronize()
    assert cnt == 8 and bool(torch.isfinite(met).all()) and bool(torch.isfinite(P).all())


def test_fused_rollout_and_recurrent_envelopes(ctx, dev):
    """Shapes outside a fused kernel's envelope are reported by the *_supported query and refused by the entry point."""
    wide_obs = mlp_desc(40, [512, 256, 128], 6, ACT_ELU, False, True)
    wide_c = mlp_desc(40, [512, 256, 128], 1, ACT_ELU, False, False)
    assert not ctx.rollout_step_supported(wide_obs, wide_c)
    ok_p, ok_c = mlp_desc(17, [512, 256, 128], 6, ACT_ELU, True, True), mlp_desc(17, [512, 256, 128], 1, ACT_ELU, True, False)
    assert ctx.rollout_step_supported(ok_p, ok_c)
    d = L.lstm_policy_desc(17, 6, lstm_hidden=128)                            # this build's recurrent kernels hold a 64-wide cell
    N = 32
    buf = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(L.RlxError, match="lstm_hidden"):
        ctx.ppo_lstm_act(d, buf(400000), ok_c, buf(200000), buf(N, 17), buf(N, 128), buf(N, 128), L.prng_key(0), buf(N, 6), None,
                         buf(N), buf(N))


def test_one_call_rollout_refuses_what_the_step_call_refuses(ctx, dev):
    """rlx_ppo_rollout_f32: NULL arrays / T = 0 are RLX_EINVAL before anything is launched, a network outside the fused kernel's
    envelope is RLX_EUNSUP from its first step (same message as rlx_ppo_rollout_step_f32), and the key is untouched."""
    T, N, O, A = 3, 64, 17, 6
    ok_p, ok_c = mlp_desc(O, [512, 256, 128], A, ACT_ELU, True, True), mlp_desc(O, [512, 256, 128], 1, ACT_ELU, True, False)
    n_p, n_c = (ctx.lib.rlx_mlp_param_count(ctypes.byref(d)) for d in (ok_p, ok_c))
    P, C = torch.zeros(n_p, device=dev), torch.zeros(n_c, device=dev)
    buf = lambda *s: torch.zeros(*s, device=dev)
    env = dict(seed=1, env_id_offset=0, t=0, horizon=10, p_term=0.0, reward_noise=0.0, final_obs=buf(T, N, O), reward=buf(T, N),
               terminated=buf(T, N), ep_step=torch.zeros(N, dtype=torch.int32, device=dev), ep_ret=buf(N), last_ret=buf(N),
               last_len=buf(N), episode_stats=buf(4))
    states, obs_last, actions, values, logps = buf(T, N, O), buf(N, O), buf(T, N, A), buf(T, N), buf(T, N)
    key = L.prng_key(3)
    k = (ctypes.c_uint32 * 2)(int(key[0]), int(key[1]))
    args = lambda st_ptr, T_: (ctx.h, ctypes.byref(ok_p), _p(P), ctypes.byref(ok_c), _p(C), st_ptr, _p(obs_last), k, 1, _p(actions),
                               _p(values), _p(logps), T_, N, 0, None, None, 0, N, 1, 0, 0, 10, 0.0, 0.0, _p(env["final_obs"]),
                               _p(env["reward"]), _p(env["terminated"]), _p(env["ep_step"]), _p(env["ep_ret"]), _p(env["last_ret"]),
                               _p(env["last_len"]), _p(env["episode_stats"]), _stream())
    assert ctx.lib.rlx_ppo_rollout_f32(*args(None, T)) != 0 and b"NULL" in ctx.lib.rlx_last_error()
    assert ctx.lib.rlx_ppo_rollout_f32(*args(_p(states), 0)) != 0 and b"bad sizes" in ctx.lib.rlx_last_error()
    assert (k[0], k[1]) == (int(key[0]), int(key[1]))
    wide_p, wide_c = mlp_desc(40, [512, 256, 128], A, ACT_ELU, False, True), mlp_desc(40, [512, 256, 128], 1, ACT_ELU, False, False)
    with pytest.raises(L.RlxError, match="envelope"):
        ctx.rollout(wide_p, buf(400000), wide_c, buf(400000), buf(T, N, 40), buf(N, 40), key, actions, values, logps,
                    dict(env, final_obs=buf(T, N, 40)))
    new_key = ctx.rollout(ok_p, P, ok_c, C, states, obs_last, key, actions, values, logps, env)      # and a valid call still works
    torch.cuda.synchronize()
    assert not np.array_equal(new_key, key) and bool(torch.isfinite(values).all())


def _ppo_plugin(nr_envs=512, nr_steps=16, minibatch=4096, epochs=1):
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
    config.environment.nr_envs = nr_envs
    config.algorithm.nr_steps, config.algorithm.minibatch_size, config.algorithm.nr_epochs = nr_steps, minibatch, epochs
    config.algorithm.total_timesteps = nr_envs * nr_steps * 4
    env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    return get_algorithm_model_class("ppo.hip")(config, env, env, "/tmp/rlx_window", None), env


@pytest.mark.parametrize("poison_at", ["before_rollout", "before_update"])
def test_a_weight_outside_the_engine_window_trains_on_the_exact_engine(poison_at):
    """gemm_bx.h: weights enter the fp16 pipe times 64, so |w| >= 1023 becomes inf in the weight image and NaN in every product.
    What must happen: training CONTINUES on the exact-fp32 engine with a logged warning, and the iteration
    meets the oracle at 1e-5.
    before_rollout: the library's check in rlx_ppo_rollout_begin finds the weight, the T acting steps run on the exact layers
      (finite actions / values / log-probs), the plugin switches the context over before the update.
    before_update: the weight is written after the rollout (the acting check has passed): the update on the split-operand engine
      comes back non-finite -- its optimizer steps were skipped on the device --, the plugin restores the state it saved in front
      of the update, switches to the exact engine and runs the same update again."""
    import logging
    from oracle import nets, ppo as oppo
    m, env = _ppo_plugin()
    assert m.ctx.get_counter("gemm_bx") == 1
    batch = m._alloc_batch()
    n_upd = m.nr_epochs * m.nr_minibatches
    met = torch.zeros(n_upd, 10, device=m.device)
    state, _ = env.reset()
    W = 17 * 512 + 3 * 512 + 5                              # W1[0][5] of the policy (512-LN-256-128 nets: after W0, b0, LN scale, LN bias)
    seen = {}
    if poison_at == "before_rollout":
        m.pparams[W] = 2000.0
    real_update, real_adv = m.update, m.compute_advantages

    def compute_advantages(b):
        real_adv(b)
        if poison_at == "before_update":                    # (after the rollout and GAE, in front of the state the plugin saves)
            m.pparams[W] = 2000.0
    m.compute_advantages = compute_advantages

    def update(b, mo):
        if "key" not in seen:
            seen["key"] = np.array(m.key, copy=True)
            seen["P"], seen["C"] = m.pparams.cpu().numpy().copy(), m.cparams.cpu().numpy().copy()
        seen["calls"] = seen.get("calls", 0) + 1
        return real_update(b, mo)
    m.update = update
    # (the "rl_x" logger may have been configured by an earlier Runner test -- own handlers, propagate off --, so the records are
    #  collected by a handler of this test on that logger itself, with the earlier handlers set aside)
    records = []

    class Collect(logging.Handler):
        def emit(self, record):
            records.append(record)
    lg = logging.getLogger("rl_x")
    old_handlers, old_level = lg.handlers[:], lg.level
    lg.handlers[:] = [Collect()]
    lg.setLevel(logging.INFO)
    try:
        state = m.train_iteration(batch, state, met)
    finally:
        lg.handlers[:] = old_handlers
        lg.setLevel(old_level)
    torch.cuda.synchronize()
    assert any(r.levelno == logging.WARNING and "exact-fp32" in r.getMessage() and "1023" in r.getMessage() for r in records)
    assert m.ctx.get_counter("gemm_bx") == 0                 # the context stays on the exact engine
    assert m.ctx.get_counter("bx_window_fallbacks") == (1 if poison_at == "before_rollout" else 0)
    assert seen["calls"] == (1 if poison_at == "before_rollout" else 2)
    assert m.opt_count == n_upd and all(np.isfinite(m.last_host_metrics[:10]))
    for x in (m.pparams, m.cparams, m.pm, m.pv, m.cm, m.cv, batch.actions, batch.values, batch.log_probs, batch.advantages):
        assert bool(torch.isfinite(x).all())
    # ---- the oracle's update from the same rollout, parameters (poisoned weight included) and key
    O, A = m.obs_dim, m.act_dim
    ps, cs = nets.make_spec("B", O, A, True), nets.make_spec("B", O, 1, False)
    assert ps.n_params == m.pparams.numel()
    f = lambda x: x.cpu().numpy()
    pst, cst = oppo.TrainState(ps, seen["P"]), oppo.TrainState(cs, seen["C"])
    cfg = dict(minibatch_size=int(m.minibatch_size), nr_epochs=int(m.nr_epochs), learning_rate=float(m.learning_rate),
               clip_range=m.clip_range, entropy_coef=m.entropy_coef, critic_coef=m.critic_coef, max_grad_norm=m.max_grad_norm)
    out, key_e, _ = oppo.update(pst, cst, f(batch.states), f(batch.actions), f(batch.advantages), f(batch.returns), None,
                                f(batch.log_probs), seen["key"], cfg, partitionable=bool(m.scheme))
    assert np.array_equal(m.key, key_e) and len(out) == n_upd
    got = met.cpu().numpy()
    for i, name in ((0, "loss/policy_gradient_loss"), (1, "loss/critic_loss")):      # first update: identical parameters on both sides
        np.testing.assert_allclose(got[0, i], out[0][name], rtol=1e-5, atol=1e-6)
    # the parameters after the iteration (two fp32 Adam trajectories: the smoke test's bar)
    for g, e in ((f(m.pparams), pst.params), (f(m.cparams), cst.params)):
        d = np.abs(g - e)
        assert (d <= 2e-5 + 1e-3 * np.abs(e)).mean() > 0.999 and d.max() <= 2 * float(m.learning_rate) * n_upd, (d.max(), (d > 2e-5).mean())
    # ---- and the next iteration simply runs (exact engine: no check, no images)
    state = m.train_iteration(batch, state, met)
    assert all(np.isfinite(m.last_host_metrics[:10])) and m.opt_count == 2 * n_upd


def test_non_finite_training_on_the_exact_engine_still_raises():
    """The fallback is for the ENGINE's window only: when the exact-fp32 engine's update is non-finite too, the run stops with the
    last finite parameters intact (a NaN observation here: no engine can train on it)."""
    m, env = _ppo_plugin()
    batch = m._alloc_batch()
    met = torch.zeros(m.nr_epochs * m.nr_minibatches, 10, device=m.device)
    state, _ = env.reset()
    before = (m.pparams.clone(), m.cparams.clone())
    real_adv = m.compute_advantages

    def poisoned(b):
        real_adv(b)
        b.advantages.fill_(float("nan"))                    # every minibatch of both networks: every optimizer step is skipped
        b.returns.fill_(float("nan"))
    m.compute_advantages = poisoned
    with pytest.raises(FloatingPointError) as e:
        m.train_iteration(batch, state, met)
    assert "SKIPPED" in str(e.value) and "exact fp32 already" in str(e.value)
    assert torch.equal(m.pparams, before[0]) and torch.equal(m.cparams, before[1])
