"""GPU: SAC kernels vs the oracle -- acting, replay gather, and one whole update (losses,
gradients via the post-Adam parameters, Polyak targets, key schedule)."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo, prng, sac
from rlx_amd.hip import RlxError, SacHparams, mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


def _descs(ps, qs):
    return (mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False),
            mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False))


def _assert_split_engine_only(ctx, B):
    """Every GEMM-shaped launch of a B >= 4096 update must have run on the split-operand (fp16-pipe) engine: an exact-fp32 row
    here is a silent engine fallback (round 4 found two of them with a kernel table: 1342 -> 1759 updates/s once fixed) -- it
    now fails a test instead of a benchmark (VERDICT r04 #4 iii)."""
    rows = ctx.prof_rows()
    if B >= 4096:
        gemm = [r for r in rows if r["kernel"] in ("k_gemm_fwd", "k_gemm_dx", "k_gemm_dw")]
        assert gemm and not any(r["engine"] == 0 for r in gemm), [r for r in gemm if r["engine"] == 0]


@pytest.mark.parametrize("O,A,B,H", [(376, 17, 256, 256), (40, 6, 100, 64), (376, 17, 4096, 256), (11, 3, 200, 64),
                                     (3, 1, 64, 64), (45, 5, 130, 128)])
@pytest.mark.parametrize("scheme,head_scale", [(1, 0.1), (0, 0.1), (1, 1.0)])
def test_sac_update_matches_oracle(ctx, dev, O, A, B, H, scheme, head_scale):
    """head_scale 0.1: std ~ 1, tanh rarely saturates -> fp32 is well conditioned, tight tolerances vs the float64
    oracle.  head_scale 1.0: raw lecun heads give std up to e^2 and many saturated actions; there
    log(1 - tanh(u)^2 + 1e-6) cancels catastrophically in ANY fp32 implementation (numpy-fp32 vs float64 differ by
    5e-4 on q_loss, 2.6e-3 on the entropy for these inputs), so only a loose 3e-3 agreement is meaningful."""
    if (scheme == 0 or head_scale == 1.0) and B > 256:
        pytest.skip("covered at small batch")
    tol = 1e-5 if head_scale < 1.0 else 3e-3
    rng = np.random.default_rng(O + B)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= head_scale
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = (qp + 0.01 * rng.standard_normal(qp.shape)).astype(np.float32)
    s = rng.standard_normal((B, O)).astype(np.float32)
    s2 = rng.standard_normal((B, O)).astype(np.float32)
    a = np.tanh(rng.standard_normal((B, A))).astype(np.float32)
    r = rng.standard_normal(B).astype(np.float32)
    term = (rng.random(B) < 0.2).astype(np.float32)
    log_alpha = np.float32(-0.3)
    gamma, tau, lr = 0.99, 0.005, 3e-4
    key = prng.prng_key(11)
    # ---- oracle: float64 math on the same fp32 inputs
    f = lambda x: x.astype(np.float64)
    new_key_e, e1, e2 = sac.sample_noise(key, B, A, bool(scheme))
    met_e, gp_e, gq_e, ga_e = sac.loss_and_grads(ps, f(pp), qs, f(qp), f(qtp), np.float64(log_alpha), f(s), f(s2), f(a), f(r),
                                                 f(term), f(e1), f(e2), gamma, -float(A))
    z = lambda g: (np.zeros_like(g), np.zeros_like(g))
    pp_e, _, _ = oppo.adam_step(f(pp), gp_e, *z(gp_e), 0, lr)
    qp_e, _, _ = oppo.adam_step(f(qp), gq_e, *z(gq_e), 0, lr)
    la_e, _, _ = oppo.adam_step(np.array([log_alpha], np.float64), np.array([ga_e]), np.zeros(1), np.zeros(1), 0, lr)
    qt_e = sac.polyak(qp_e, f(qtp), tau)
    # ---- HIP
    pd, qd = _descs(ps, qs)
    P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qtp, dev)
    LA = _t(np.array([log_alpha]), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    hp = SacHparams(gamma, tau, -float(A), -20.0, 2.0, lr, lr, lr, 0.9, 0.999, 1e-8)
    met = torch.zeros(10, device=dev)
    ctx.prof_begin()
    new_key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av,
                                  (_t(s, dev), _t(s2, dev), _t(a, dev), _t(r, dev), _t(term, dev)), key, 0, hp, met, scheme)
    ctx.prof_end()
    _assert_split_engine_only(ctx, B)
    assert np.array_equal(new_key, new_key_e) and cnt == 1
    m = met.cpu().numpy()
    names = ["loss/q_loss", "loss/policy_loss", "loss/entropy_loss", "entropy/entropy", "entropy/alpha", "q_value/q_value"]
    for i, n in enumerate(names):
        assert m[i] == pytest.approx(float(met_e[n]), rel=tol, abs=tol), n
    ntol = 1e-5 if head_scale < 1.0 else 0.06
    assert m[6] == pytest.approx(np.linalg.norm(gp_e), rel=ntol)
    assert m[7] == pytest.approx(np.linalg.norm(gq_e), rel=ntol)
    assert m[8] == pytest.approx(abs(float(ga_e)), rel=ntol, abs=tol)
    # first Adam step is -lr*sign(g) for |g| >> eps: compare the parameters (robust subset) and the moments (= gradients)
    rp = np.linalg.norm(pm.cpu().numpy() * 10 - gp_e) / np.linalg.norm(gp_e)
    rq = np.linalg.norm(qm.cpu().numpy() * 10 - gq_e) / np.linalg.norm(gq_e)
    print(f"sac update O={O} A={A} B={B} H={H} scheme={scheme} head={head_scale}: ||dg||/||g|| policy {rp:.2e} critic {rq:.2e}; "
          + " ".join(f"{n.split('/')[1]} {abs(m[i] - float(met_e[n])) / max(abs(float(met_e[n])), 1e-30):.1e}" for i, n in enumerate(names)))
    assert rp < (1e-5 if head_scale < 1 else 0.05)
    assert rq < (1e-5 if head_scale < 1 else 0.01)
    assert LA.item() == pytest.approx(la_e[0], abs=1e-6)
    d = np.abs(P.cpu().numpy() - pp_e)
    assert (d <= 2e-5).mean() > (0.99 if head_scale < 1 else 0.9) and d.max() <= 2 * lr + 1e-6
    np.testing.assert_allclose(QT.cpu().numpy(), qt_e, rtol=1e-4, atol=2 * lr * tau + 1e-6)   # Polyak of the updated critics


def test_sac_act_and_replay_gather(ctx, dev):
    O, A, N = 376, 17, 96
    rng = np.random.default_rng(0)
    ps, qs = sac.make_specs(O, A, 256)
    pp = sac.lecun_normal_init(ps, rng)
    pd, _ = _descs(ps, qs)
    obs = rng.standard_normal((N, O)).astype(np.float32)
    key = prng.prng_key(2)
    act = torch.empty(N, A, device=dev)
    new_key = ctx.sac_act(pd, _t(pp, dev), _t(obs, dev), key, act, -20.0, 2.0)
    ks = prng.split(key, 2)
    mean, ls, _, _ = sac.policy_forward(ps, pp.astype(np.float64), obs.astype(np.float64), -20.0, 2.0)
    exp = np.tanh(mean + np.exp(ls) * prng.normal(ks[1], (N, A)))
    assert np.array_equal(new_key, ks[0])
    np.testing.assert_allclose(act.cpu().numpy(), exp, rtol=1e-5, atol=5e-6)
    det = torch.empty(N, A, device=dev)
    k2 = ctx.sac_act(pd, _t(pp, dev), _t(obs, dev), key, det, -20.0, 2.0, deterministic=True)
    assert np.array_equal(k2, key)
    np.testing.assert_allclose(det.cpu().numpy(), np.tanh(mean), rtol=1e-5, atol=5e-6)
    # the env-facing action from the same launch (get_processed_action, sac/flax/policy.py:44-48)
    low = rng.uniform(-2.0, -0.5, A).astype(np.float32)
    high = rng.uniform(0.5, 3.0, A).astype(np.float32)
    act2, proc = torch.empty(N, A, device=dev), torch.empty(N, A, device=dev)
    k3 = ctx.sac_act(pd, _t(pp, dev), _t(obs, dev), key, act2, -20.0, 2.0, processed=(_t(low, dev), _t(0.5 * (high - low), dev), proc))
    assert np.array_equal(k3, ks[0]) and torch.equal(act2, act)
    a64 = act.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(proc.cpu().numpy(), sac.processed_action(a64, low.astype(np.float64), high.astype(np.float64)),
                               rtol=1e-6, atol=1e-6)
    # replay ring gather
    cap, NE = 7, 5
    rb = sac.ReplayBuffer(cap * NE, NE, O, A, np.random.default_rng(1))
    for t in range(9):
        rb.add(rng.standard_normal((NE, O)), rng.standard_normal((NE, O)), rng.standard_normal((NE, A)),
               rng.standard_normal(NE), (rng.random(NE) < 0.5))
    i1, i2 = rb.sample_indices(64)
    ring = tuple(_t(x, dev) for x in (rb.states, rb.next_states, rb.actions, rb.rewards, rb.terminations))
    out = (torch.empty(64, O, device=dev), torch.empty(64, O, device=dev), torch.empty(64, A, device=dev),
           torch.empty(64, device=dev), torch.empty(64, device=dev))
    ctx.sac_replay_sample(ring, _t(i1, dev, np.int32), _t(i2, dev, np.int32), out)
    for got, exp in zip(out, rb.gather(i1, i2)):
        assert np.array_equal(got.cpu().numpy(), exp.astype(np.float32))


@pytest.mark.parametrize("O,A", [(3, 1), (45, 5)])
def test_sac_act_odd_observation_widths(ctx, dev, O, A):
    """get_action for observation widths the 16-B vector loads do not like (narrow; wide but not a multiple of 4)."""
    import torch
    from oracle import nets, sac as osac
    rng = np.random.default_rng(O)
    ps, _ = osac.make_specs(O, A, 64)
    pp = osac.lecun_normal_init(ps, rng)
    obs = rng.standard_normal((50, O)).astype(np.float32)
    out, _ = nets.forward(ps, pp.astype(np.float64), obs.astype(np.float64))
    exp = np.tanh(out[:, :A])
    from rlx_amd.hip import mlp_desc
    d = mlp_desc(O, [64, 64], 2 * A, nets.ACT_RELU, False, False)
    action = torch.empty(50, A, device=dev)
    ctx.sac_act(d, torch.from_numpy(pp).to(dev), torch.from_numpy(obs).to(dev), np.array([1, 2], np.uint32), action, -20.0, 2.0,
                deterministic=True)
    np.testing.assert_allclose(action.cpu().numpy(), exp, rtol=1e-5, atol=2e-6)


def test_golden_sac_fixture(ctx, dev):
    """HIP vs the committed golden vectors (tests/golden/sac.npz): losses, gradients (= 10 x first Adam moment), key."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sac.npz"))
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    ps, qs = sac.make_specs(O, A, H)
    pd, qd = _descs(ps, qs)
    P, Q, QT = _t(g["pparams"], dev), _t(g["qparams"], dev), _t(g["qtarget"], dev)
    LA = _t(np.array([g["log_alpha"]], np.float32), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    hp = SacHparams(float(g["gamma"]), 0.005, float(g["target_entropy"]), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8)
    met = torch.zeros(10, device=dev)
    batch = tuple(_t(g[k], dev) for k in ("states", "next_states", "actions", "rewards", "terminations"))
    new_key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, g["key"], 0, hp, met, 1)
    assert np.array_equal(new_key, g["new_key"]) and cnt == 1
    m = met.cpu().numpy()
    for i, n in enumerate(("q_loss", "policy_loss", "entropy_loss", "entropy", "alpha", "q_value")):
        assert m[i] == pytest.approx(float(g[n]), rel=1e-5, abs=1e-5), n
    assert np.linalg.norm(pm.cpu().numpy() * 10 - g["gpolicy"]) / np.linalg.norm(g["gpolicy"]) < 1e-5
    assert np.linalg.norm(qm.cpu().numpy() * 10 - g["gcritic"]) / np.linalg.norm(g["gcritic"]) < 1e-5
    assert am.item() * 10 == pytest.approx(float(g["g_log_alpha"]), rel=1e-4)


@pytest.mark.parametrize("O,A,B", [(376, 17, 4096), (40, 6, 100), (11, 3, 200), (3, 1, 64)])
def test_sac_full_jit_update_matches_oracle(ctx, dev, O, A, B):
    """The fully jitted flavour (sac/flax_full_jit): 512-LayerNorm-256-128 ELU policy and twin critics (wide first layer on
    the GEMM kernels + k_ln_act, narrow ones on the fused first-layer kernels), key schedule split(key, 2B+2) with the noise
    keys in two contiguous blocks, device-side replay index draw from keys[1].  BASELINE.json configs[3] shape included."""
    rng = np.random.default_rng(O * 7 + B)
    ps, qs = sac.make_specs(O, A, arch="full_jit")
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = (qp + 0.01 * rng.standard_normal(qp.shape)).astype(np.float32)
    s = rng.standard_normal((B, O)).astype(np.float32)
    s2 = rng.standard_normal((B, O)).astype(np.float32)
    a = np.tanh(rng.standard_normal((B, A))).astype(np.float32)
    r = rng.standard_normal(B).astype(np.float32)
    term = (rng.random(B) < 0.2).astype(np.float32)
    log_alpha = np.float32(-0.3)
    gamma, tau, lr = 0.99, 0.005, 3e-4
    key = prng.prng_key(13)
    f = lambda x: x.astype(np.float64)
    new_key_e, e1, e2 = sac.sample_noise(key, B, A, True, schedule=1)
    met_e, gp_e, gq_e, ga_e = sac.loss_and_grads(ps, f(pp), qs, f(qp), f(qtp), np.float64(log_alpha), f(s), f(s2), f(a), f(r),
                                                 f(term), f(e1), f(e2), gamma, -float(A))
    pd, qd = _descs(ps, qs)
    P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qtp, dev)
    LA = _t(np.array([log_alpha]), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    hp = SacHparams(gamma, tau, -float(A), -20.0, 2.0, lr, lr, lr, 0.9, 0.999, 1e-8)
    hp.key_schedule = 1
    met = torch.zeros(10, device=dev)
    # replay index draw of this update (same key), against the oracle's restatement of jax.random.randint
    idx1, idx2 = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    ctx.sac_replay_draw(key, B, 244, 4096, idx1, idx2)
    i1_e, i2_e = sac.replay_indices(key, B, 244, 4096)
    assert np.array_equal(idx1.cpu().numpy(), i1_e) and np.array_equal(idx2.cpu().numpy(), i2_e)
    ctx.prof_begin()
    new_key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av,
                                  (_t(s, dev), _t(s2, dev), _t(a, dev), _t(r, dev), _t(term, dev)), key, 0, hp, met, 1)
    ctx.prof_end()
    _assert_split_engine_only(ctx, B)
    assert np.array_equal(new_key, new_key_e) and cnt == 1
    m = met.cpu().numpy()
    names = ["loss/q_loss", "loss/policy_loss", "loss/entropy_loss", "entropy/entropy", "entropy/alpha", "q_value/q_value"]
    for i, n in enumerate(names):
        assert m[i] == pytest.approx(float(met_e[n]), rel=1e-5, abs=1e-5), n
    rp = np.linalg.norm(pm.cpu().numpy() * 10 - gp_e) / np.linalg.norm(gp_e)
    rq = np.linalg.norm(qm.cpu().numpy() * 10 - gq_e) / np.linalg.norm(gq_e)
    print(f"sac full_jit update O={O} A={A} B={B}: ||dg||/||g|| policy {rp:.2e} critic {rq:.2e}")
    assert rp < 1e-5 and rq < 1e-5
    assert am.item() * 10 == pytest.approx(float(ga_e), rel=1e-4, abs=1e-7)
    # forward-only entry on the same nets (acting): deterministic action = tanh(mean)
    act = torch.empty(B, A, device=dev)
    ctx.sac_act(pd, _t(pp, dev), _t(s, dev), key, act, -20.0, 2.0, deterministic=True)
    mean, _, _, _ = sac.policy_forward(ps, f(pp), f(s), -20.0, 2.0)
    np.testing.assert_allclose(act.cpu().numpy(), np.tanh(mean), rtol=1e-5, atol=5e-6)


def test_direct_replay_writes_equal_the_generic_path(dev):
    """sac.hip's per-step code writes the transition straight into the replay ring slot when the env offers `step_into`
    (acting kernel -> action row, env kernel -> final observation / reward / termination rows).  Same ring contents, bit for
    bit, as the generic path (env.step + five copies), including the processed action the env receives."""
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.sac.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    rings = []
    for direct in (True, False):
        config = ConfigDict()
        config.runner = runner_cfg("train")
        config.algorithm = get_algorithm_config("sac.hip")
        config.environment = get_environment_config("synthetic.random_obs")
        config.environment.nr_envs, config.environment.obs_dim, config.environment.act_dim = 64, 40, 5
        config.environment.horizon, config.environment.termination_probability = 7, 0.05
        config.algorithm.batch_size, config.algorithm.buffer_size = 32, 64 * 16
        env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
        m = get_algorithm_model_class("sac.hip")(config, env, env, "/tmp/rlx_dr", None)
        m.direct_replay = direct
        m._alloc()
        state, _ = env.reset()
        state = state.clone()
        gen = torch.Generator(device=m.device)
        gen.manual_seed(3)
        for i in range(20):                      # wraps the 16-slot ring; warm-up (uniform) and policy actions
            state = m.vector_step(env, state, warmup=i < 5, gen=gen)
        torch.cuda.synchronize()
        rings.append([x.clone() for x in m.ring] + [state.clone()])
    for a, b in zip(*rings):
        assert torch.equal(a, b)


@pytest.mark.parametrize("O,A,B,H", [(376, 17, 4096, 256), (11, 3, 200, 64)])
def test_sac_update_one_and_two_chain_layouts_are_bit_identical(ctx, dev, O, A, B, H):
    """The update issued on one stream and as two concurrent chains (per-call key and Adam schedule read from device
    memory): same kernels on the same buffers in the same per-buffer order -> identical parameters, moments, targets,
    metrics and keys after six consecutive updates."""
    rng = np.random.default_rng(O * 7 + B)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))),
            rng.standard_normal(B), (rng.random(B) < 0.2)]
    pd, qd = _descs(ps, qs)
    hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 1e-3, 2e-4, 0.9, 0.999, 1e-8)
    results = []
    try:
        for two in (0, 1, 1):
            ctx.set_option("two_streams", two)
            P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
            LA = _t(np.array([-0.3]), dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            met = torch.zeros(10, device=dev)
            batch = tuple(_t(x, dev) for x in data)
            key, cnt, mets = prng.prng_key(4), 0, []
            for _ in range(6):
                key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
                mets.append(met.clone())
            torch.cuda.synchronize()
            results.append([np.asarray(key), np.int64(cnt)] + [x.cpu().numpy() for x in (P, pm, pv, Q, qm, qv, QT, LA, am, av)]
                           + [torch.stack(mets).cpu().numpy()])
    finally:
        ctx.set_option("two_streams", 1)
    assert np.isfinite(results[0][-1]).all() and results[0][1] == 6
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("O,A,B,H", [(376, 17, 4096, 256), (11, 3, 200, 64), (40, 5, 333, 64)])
def test_sac_update_from_the_ring_equals_sample_then_update(ctx, dev, O, A, B, H):
    """rlx_sac_hparams::ring_*: the update gathers the sampled transitions itself (into the batch arguments) in the launch that
    lays out the critics' input rows.  Batch arrays, parameters, moments, targets, metrics and key are bit-identical to
    rlx_sac_replay_sample_f32 followed by the plain update (wide 16-byte-aligned rows, a narrow ragged and a wide ragged width)."""
    rng = np.random.default_rng(O * 13 + B)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    CAP, N = 6, 50
    ring_np = [rng.standard_normal((CAP, N, O)), rng.standard_normal((CAP, N, O)), np.tanh(rng.standard_normal((CAP, N, A))),
               rng.standard_normal((CAP, N)), (rng.random((CAP, N)) < 0.2)]
    ring = tuple(_t(x, dev) for x in ring_np)
    pd, qd = _descs(ps, qs)
    results = []
    # third mode: the ring source with states / next_states = NULL (the caller does not want the gathered observation rows back;
    # wide observations only -- the narrow case is an error, below)
    for mode in ("sample", "ring", "ring_no_states")[:3 if O > 32 else 2]:
        P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
        LA = _t(np.array([-0.3]), dev)
        pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
        am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        met = torch.zeros(10, device=dev)
        batch = tuple(torch.full(sh, 7.0, device=dev) for sh in ((B, O), (B, O), (B, A), (B,), (B,)))
        key, cnt, mets, batches = prng.prng_key(4), 0, [], []
        irng = np.random.default_rng(9)
        for _ in range(3):
            i1 = torch.from_numpy(irng.integers(0, CAP, B).astype(np.int32)).to(dev)
            i2 = torch.from_numpy(irng.integers(0, N, B).astype(np.int32)).to(dev)
            hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 1e-3, 2e-4, 0.9, 0.999, 1e-8)
            if mode != "sample":
                (hp.ring_states, hp.ring_next_states, hp.ring_actions, hp.ring_rewards,
                 hp.ring_terminations) = (x.data_ptr() for x in ring)
                hp.ring_idx1, hp.ring_idx2, hp.ring_nr_envs = i1.data_ptr(), i2.data_ptr(), N
            else:
                ctx.sac_replay_sample(ring, i1, i2, batch)
            passed = (None, None) + batch[2:] if mode == "ring_no_states" else batch
            key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, passed, key, cnt, hp, met, 1)
            mets.append(met.clone())
            batches += [x.clone() for x in batch]
        torch.cuda.synchronize()
        results.append([np.asarray(key), np.int64(cnt)] + [x.cpu().numpy() for x in [P, pm, pv, Q, qm, qv, QT, LA, am, av] + batches]
                       + [torch.stack(mets).cpu().numpy()])
    assert np.isfinite(results[0][-1]).all() and results[0][1] == 3
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    exp = ring_np[2].astype(np.float32)[i1.cpu().numpy(), i2.cpu().numpy()]
    assert np.array_equal(results[1][12 + 3 * 5 - 3], exp)           # the last gathered action rows, against numpy fancy indexing
    if O > 32:      # without the observation arrays: every output the caller DID request is bit-identical, the arrays it did not pass are untouched
        for j, (a, b) in enumerate(zip(results[1], results[2])):
            in_batches = 12 <= j < 12 + 15
            if in_batches and (j - 12) % 5 < 2:
                assert (b == 7.0).all(), j
            else:
                assert np.array_equal(a, b), j
    else:           # narrow observations are read by the policy from these arrays: refusing NULL beats reading garbage
        hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 1e-3, 2e-4, 0.9, 0.999, 1e-8)
        (hp.ring_states, hp.ring_next_states, hp.ring_actions, hp.ring_rewards, hp.ring_terminations) = (x.data_ptr() for x in ring)
        hp.ring_idx1, hp.ring_idx2, hp.ring_nr_envs = i1.data_ptr(), i2.data_ptr(), N
        with pytest.raises(RlxError, match="obs_dim > 32"):
            ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, (None, None) + batch[2:], key, cnt, hp, met, 1)
    with pytest.raises(RlxError, match="ring source"):      # NULL observation arrays without a ring to gather from
        hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 1e-3, 2e-4, 0.9, 0.999, 1e-8)
        ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, (None, None) + batch[2:], key, cnt, hp, met, 1)


@pytest.mark.parametrize("arch", ["flax", "full_jit"])
def test_sac_twin_critic_launches_match_the_sequential_passes(ctx, dev, arch):
    """Both critics of a pair in one launch per layer (grid.y = 2) against the two sequential passes: identical forward
    values and input gradients (same kernels, same tiles per net) -> identical losses / policy gradient of the first
    update; the critics' weight gradients are summed over half as many M-slabs (fp32 order), so their moments and the
    later updates agree to rounding."""
    O, A, B, H = 376, 17, 4096, 256
    rng = np.random.default_rng(5)
    ps, qs = sac.make_specs(O, A, H, arch=arch)        # full_jit: LayerNorm first layer, three hidden layers
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))),
            rng.standard_normal(B), (rng.random(B) < 0.2)]
    pd, qd = _descs(ps, qs)
    hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, int(arch == "full_jit"))
    res = []
    try:
        for twin in (1, 0):
            ctx.set_option("sac_twin", twin)
            P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
            LA = _t(np.array([-0.3]), dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            met = torch.zeros(10, device=dev)
            batch = tuple(_t(x, dev) for x in data)
            key, cnt = prng.prng_key(4), 0
            key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            first = [met.cpu().numpy().copy(), pm.cpu().numpy().copy(), qm.cpu().numpy().copy()]
            for _ in range(3):
                key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            res.append(first + [met.cpu().numpy().copy(), P.cpu().numpy(), Q.cpu().numpy()])
    finally:
        ctx.set_option("sac_twin", 1)
    t, s = res
    assert np.array_equal(t[0][:7], s[0][:7]) and np.array_equal(t[1], s[1])      # losses, policy gradient: bit for bit
    assert np.linalg.norm(t[2] - s[2]) / np.linalg.norm(s[2]) < 2e-7              # critic gradients: summation order only
    assert t[0][7] == pytest.approx(s[0][7], rel=1e-6)
    np.testing.assert_allclose(t[3][:6], s[3][:6], rtol=2e-4, atol=1e-5)           # after four updates
    assert np.abs(t[4] - s[4]).max() < 1e-4 and np.abs(t[5] - s[5]).max() < 1e-4


def test_predrawn_replay_indices_keep_the_reference_stream(dev):
    """sac.hip draws the replay indices of the next 32 updates at once (one pinned H2D copy instead of two per step).  Whatever
    the pattern of ring sizes the updates actually find -- growing by one per step, constant once full, several updates at one
    size, an evaluation pause -- the indices are those of the reference's lazy draws (numpy PCG64, replay_buffer.py:31-32)."""
    from test_gpu_obs_indices import _plugin
    cls, config, env = _plugin("sac.hip", dict(nr_envs=16, obs_dim=8, act_dim=2),
                               dict(batch_size=48, buffer_size=16 * 40, learning_starts=16, total_timesteps=16 * 4), None, None)
    m = cls(config, env, env, "/tmp/rlx_predraw", None)
    m._alloc()
    lazy = np.random.default_rng(int(m.seed))
    sizes = (list(range(1, 12)) + [12, 12, 12] + list(range(13, 41)) + [40] * 50 + [7, 8, 9] + [40] * 5 +
             [sz for sz in range(1, 30) for _ in (0, 1)] + list(range(1, 40, 3)))   # two updates per step; three steps per update
    drawn = {"n": 0}
    integers = m.rng.integers

    class Counting:
        """m.rng with its integers() counted (one call = one index vector of one update)"""
        bit_generator = m.rng.bit_generator

        @staticmethod
        def integers(*a, **k):
            drawn["n"] += 1
            return integers(*a, **k)
    m.rng = Counting
    for step, size in enumerate(sizes):
        m.size = size
        i1, i2 = m._host_indices(48, 16)
        e1, e2 = lazy.integers(size, size=48), lazy.integers(16, size=48)
        assert m.consumed_rng_state() == lazy.bit_generator.state, step     # what a checkpoint has to store
        torch.cuda.synchronize()
        assert np.array_equal(i1.cpu().numpy(), e1) and np.array_equal(i2.cpu().numpy(), e2), step
    # a cadence the predictor cannot follow costs a bounded number of redraws: block length halves on a miss
    assert drawn["n"] <= 2 * 4 * len(sizes), (drawn["n"], len(sizes))
    # and the generator itself: after a rewind-free tail both are at most one block apart -- draw once more at a NEW size
    m.size = 23
    i1, _ = m._host_indices(48, 16)
    assert np.array_equal(i1.cpu().numpy(), lazy.integers(23, size=48))


def test_kept_weight_images_give_the_same_bits_as_fresh_ones(ctx, dev):
    """rlx_sac_hparams.keep_images: the networks' split weight images persist between calls and k_sac_optimizers rewrites the entries of
    every parameter / Polyak target it stores.  Four act + update rounds at B = 4096: actions, parameters, targets, moments and
    metrics are bit-identical to the rounds that lay the images out again in every call -- and no k_bx_wfrag launch is left after
    the first update (profiler rows cannot see that kernel; the counter of registered-without-launch calls does)."""
    O, A, B, H = 376, 17, 4096, 256
    rng = np.random.default_rng(9)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))),
            rng.standard_normal(B), (rng.random(B) < 0.2)]
    obs = rng.standard_normal((B, O))
    pd, qd = _descs(ps, qs)
    hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, 0)
    res = []
    try:
        for keep in (1, 0):
            hp.keep_images = keep
            P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
            LA = _t(np.array([-0.3]), dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            met = torch.zeros(10, device=dev)
            batch = tuple(_t(x, dev) for x in data)
            ob = _t(obs, dev)
            key, akey, cnt = prng.prng_key(4), prng.prng_key(5), 0
            out = []
            for _ in range(4):
                act = torch.empty(B, A, device=dev)
                akey = ctx.sac_act(pd, P, ob, akey, act, -20.0, 2.0)
                key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
                out += [act.cpu().numpy(), met.cpu().numpy().copy()]
            res.append(out + [x.cpu().numpy() for x in (P, Q, QT, pm, qm, pv, qv, LA)])
    finally:
        ctx.sac_invalidate_images()
    assert np.isfinite(res[0][-7]).all()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert not np.array_equal(res[0][0], res[0][2])      # the acting policy did change between the rounds


@pytest.mark.parametrize("writer", ["outside_plus_invalidate", "library_adam_step"])
def test_kept_weight_images_are_dropped_when_the_parameters_are_written(ctx, dev, writer):
    """The keep_images contract (include/rlx_hip.h): parameters written from outside rlx_sac_update_f32 need
    rlx_sac_invalidate_images; a LIBRARY entry point writing into one of the vectors (rlx_clip_adam_step_f32 here) drops the
    images by itself.  Either way the following act + update equal, bit for bit, the calls that never kept images."""
    O, A, B, H = 376, 17, 4096, 256
    rng = np.random.default_rng(19)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))),
            rng.standard_normal(B), (rng.random(B) < 0.2)]
    obs = rng.standard_normal((B, O))
    fake_grad = (0.01 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pd, qd = _descs(ps, qs)
    hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, 0)
    res = []
    try:
        for keep in (1, 0):
            hp.keep_images = keep
            P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
            LA = _t(np.array([-0.3]), dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            met = torch.zeros(10, device=dev)
            batch = tuple(_t(x, dev) for x in data)
            ob = _t(obs, dev)
            key, akey, cnt = prng.prng_key(4), prng.prng_key(5), 0
            key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            if writer == "library_adam_step":          # no explicit invalidate: the library sees its own write into P
                m2, v2 = torch.zeros_like(P), torch.zeros_like(P)
                ctx.clip_adam_step(P, _t(fake_grad, dev), m2, v2, 1, 1e-2, 0.5)
            else:
                P.mul_(0.5)
                QT.add_(0.01)
                ctx.sac_invalidate_images()
            act = torch.empty(B, A, device=dev)
            akey = ctx.sac_act(pd, P, ob, akey, act, -20.0, 2.0)
            key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            res.append([act.cpu().numpy(), met.cpu().numpy().copy()] + [x.cpu().numpy() for x in (P, Q, QT, pm, qm, pv, qv, LA)])
    finally:
        ctx.sac_invalidate_images()
    assert np.isfinite(res[0][1]).all()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
