"""GPU: FastSAC's networks and update steps (fastsac.hip) against the reference's own outputs (tests/golden/reference_fastsac.npz:
modules and closures of rl_x/algorithms/fastsac/pytorch executed in float64 on float32-representable inputs) and against the
float64 oracle (oracle/fastsac.py) at a second, larger shape.  Tolerances: 1e-5 relative (L2 per vector), scalars 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import fastsac as ofs
from rlx_amd.hip import FastSacHparams, lnmlp_desc

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "reference_fastsac.npz")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(dev)


def _hp(h, nr_atoms, clipped):
    hp = FastSacHparams()
    for k in ("gamma", "tau", "v_min", "v_max", "log_std_min", "log_std_max", "target_entropy", "weight_decay"):
        setattr(hp, k, float(h[k]))
    hp.lr_policy = hp.lr_critic = hp.lr_alpha = float(h["learning_rate"])
    hp.adam_b1, hp.adam_b2, hp.adam_eps = float(h["adam_beta1"]), float(h["adam_beta2"]), 1e-8
    hp.nr_atoms, hp.clipped_double_q = int(nr_atoms), int(bool(clipped))
    hp.max_grad_norm = float(h.get("max_grad_norm", -1.0))      # fixture case 2: 0.05 (torch clip_grad_norm_ active in both steps)
    return hp


def _fixture_case(c):
    z = np.load(FIX)
    k = "c%d_" % c
    g = lambda n: z[k + n]
    h = {n: float(g(n)) for n in ("gamma", "tau", "v_min", "v_max", "log_std_min", "log_std_max", "learning_rate", "weight_decay",
                                  "adam_beta1", "adam_beta2", "target_entropy", "log_alpha", "max_grad_norm")}
    O, A, NA, B = int(g("obs_dim")), int(g("act_dim")), int(g("nr_atoms")), int(g("batch"))
    pflat, qflat = ofs.make_params(int(g("param_seed")), O, A, NA)
    return z, g, h, O, A, NA, B, pflat, qflat, bool(int(g("clipped")))


def _check_sampled(z, key, full, rtol):
    idx, val, norm = z[key + "_idx"], z[key + "_val"], float(z[key + "_norm"])
    full = np.asarray(full, dtype=np.float64)
    assert np.linalg.norm(full) == pytest.approx(norm, rel=rtol)
    assert np.linalg.norm(full[idx] - val) <= rtol * np.linalg.norm(val), key


@pytest.mark.parametrize("c", [0, 1])
def test_networks_and_acting_match_the_reference_modules(ctx, dev, c):
    z, g, h, O, A, NA, B, pflat, qflat, clipped = _fixture_case(c)
    pd, qd = lnmlp_desc(O, ofs.POLICY_HIDDEN, 2 * A), lnmlp_desc(O + A, ofs.CRITIC_HIDDEN, NA)
    assert ctx.lnmlp_param_count(pd) == pflat.size and ctx.lnmlp_param_count(qd) == qflat[0].size
    s, a = _t(g("states"), dev), _t(g("actions"), dev)
    head = ctx.lnmlp_fwd(pd, _t(pflat, dev), s, torch.empty(B, 2 * A, device=dev)).cpu().numpy()
    np.testing.assert_allclose(head[:, :A], g("mean"), rtol=2e-5, atol=2e-6)
    ls = h["log_std_min"] + 0.5 * (h["log_std_max"] - h["log_std_min"]) * (np.tanh(head[:, A:].astype(np.float64)) + 1.0)
    np.testing.assert_allclose(ls, g("log_std"), rtol=2e-5, atol=2e-6)
    for k, name in ((0, "q1_logits"), (1, "q2_logits")):
        lg = ctx.lnmlp_fwd(qd, _t(qflat[k], dev), torch.cat([s, a], 1).contiguous(), torch.empty(B, NA, device=dev)).cpu().numpy()
        np.testing.assert_allclose(lg, g(name), rtol=2e-5, atol=2e-5)
    hp = _hp(h, NA, clipped)
    act = torch.empty(B, A, device=dev)
    from rlx_amd.hip import lib as L
    ctx.fastsac_act(pd, _t(pflat, dev), s, _t(g("action_scale"), dev), L.prng_key(3), act, hp, deterministic=True)
    np.testing.assert_allclose(act.cpu().numpy(), g("deterministic_action"), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("c", [0, 1, 2])
def test_critic_and_policy_steps_match_the_reference_closures(ctx, dev, c):
    from rlx_amd.hip import lib as L
    z, g, h, O, A, NA, B, pflat, qflat, clipped = _fixture_case(c)
    pd, qd = lnmlp_desc(O, ofs.POLICY_HIDDEN, 2 * A), lnmlp_desc(O + A, ofs.CRITIC_HIDDEN, NA)
    hp = _hp(h, NA, clipped)
    P, Q, QT = _t(pflat, dev), _t(np.concatenate(qflat[:2]), dev), _t(np.concatenate(qflat[2:]), dev)
    zl = torch.zeros_like
    qm, qv, pm, pv = zl(Q), zl(Q), zl(P), zl(P)
    la, am, av = _t([np.float32(h["log_alpha"])], dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    batch = tuple(_t(g(n), dev) for n in ("states", "next_states", "actions", "rewards", "dones", "truncations", "n_steps"))
    scale = _t(g("action_scale"), dev)
    met = torch.zeros(8, device=dev)
    e_next, e_cur = _t(g("noise_next"), dev), _t(g("noise_cur"), dev)
    ctx.dbg_set_sac_noise(e_next, e_cur)
    try:
        key, cnt = ctx.fastsac_critic_update(pd, P, qd, Q, qm, qv, QT, la, am, av, batch, scale, L.prng_key(5), 0, hp, met)
        assert cnt == 1
        m = met.cpu().numpy().astype(np.float64)
        for i, name in ((0, "q_loss"), (1, "entropy_loss"), (2, "q_min"), (3, "q_max"), (4, "entropy"), (5, "critic_grad_norm"),
                        (6, "entropy_grad_norm")):
            assert m[i] == pytest.approx(float(g(name)), rel=2e-5, abs=2e-6), name
        assert m[7] == pytest.approx(np.exp(np.float32(h["log_alpha"])), rel=1e-6)
        # first AdamW step from zero moments: m = (1 - b1) g  ->  the gradient is qm / (1 - b1)
        _check_sampled(z, "c%d_gcritic" % c, qm.cpu().numpy().astype(np.float64) / (1.0 - h["adam_beta1"]), 1e-5)
        _check_sampled(z, "c%d_qparams_after" % c, Q.cpu().numpy(), 1e-6)
        _check_sampled(z, "c%d_qtarget_after" % c, QT.cpu().numpy(), 1e-6)
        assert float(la[0]) == pytest.approx(float(g("log_alpha_after")), rel=1e-6)
        # the policy step runs against the updated critics and entropy coefficient
        pmet = torch.zeros(3, device=dev)
        key, pcnt = ctx.fastsac_policy_update(pd, P, pm, pv, qd, Q, la, batch[0], scale, key, 0, hp, pmet)
        assert pcnt == 1
        pmv = pmet.cpu().numpy().astype(np.float64)
        assert pmv[0] == pytest.approx(float(g("policy_loss")), rel=2e-5, abs=2e-6)
        assert pmv[1] == pytest.approx(float(g("alpha_at_policy_step")), rel=1e-5)
        assert pmv[2] == pytest.approx(float(g("policy_grad_norm")), rel=2e-5)
        _check_sampled(z, "c%d_gpolicy" % c, pm.cpu().numpy().astype(np.float64) / (1.0 - h["adam_beta1"]), 2e-5)
        _check_sampled(z, "c%d_pparams_after" % c, P.cpu().numpy(), 1e-6)
    finally:
        ctx.dbg_set_sac_noise(None, None)


def test_steps_at_the_default_batch_against_the_float64_oracle(ctx, dev):
    """B = 8192 (fastsac/pytorch/default_config.py:17), obs 48 / act 12, nr_atoms 101: the weight-gradient kernels run on the
    split-operand engine at this size; second optimizer step (non-zero moments)."""
    from rlx_amd.hip import lib as L
    rng = np.random.default_rng(2)
    O, A, NA, B = 48, 12, 101, 8192
    h = dict(gamma=0.97, tau=0.125, v_min=-20.0, v_max=20.0, log_std_min=-5.0, log_std_max=0.0, learning_rate=3e-4, weight_decay=0.001,
             adam_beta1=0.9, adam_beta2=0.95, target_entropy=0.0, log_alpha=float(np.log(0.05)))
    pflat, qflat = ofs.make_params(31, O, A, NA)
    scale = np.linspace(0.5, 1.5, A).astype(np.float32)
    f32 = lambda x: np.asarray(x, dtype=np.float32)
    s, s2 = f32(rng.standard_normal((B, O))), f32(rng.standard_normal((B, O)))
    a = f32(np.tanh(rng.standard_normal((B, A))) * scale)
    rew, done = f32(3.0 * rng.standard_normal(B)), f32(rng.random(B) < 0.2)
    trunc, nst = f32((rng.random(B) < 0.5) * done), f32(rng.integers(1, 4, B))
    e1, e2 = f32(rng.standard_normal((B, A))), f32(rng.standard_normal((B, A)))
    batch64 = tuple(np.asarray(x, dtype=np.float64) for x in (s, s2, a, rew, done, trunc, nst))
    la = float(np.float32(h["log_alpha"]))
    r = ofs.critic_step(pflat.astype(np.float64), *(q.astype(np.float64) for q in qflat), la, O, A, NA, batch64, e1, scale, h, False)
    pd, qd = lnmlp_desc(O, ofs.POLICY_HIDDEN, 2 * A), lnmlp_desc(O + A, ofs.CRITIC_HIDDEN, NA)
    hp = _hp(h, NA, False)
    P, Q, QT = _t(pflat, dev), _t(np.concatenate(qflat[:2]), dev), _t(np.concatenate(qflat[2:]), dev)
    zl = torch.zeros_like
    qm, qv, pm, pv = zl(Q), zl(Q), zl(P), zl(P)
    lad, am, av = _t([la], dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    met, pmet = torch.zeros(8, device=dev), torch.zeros(3, device=dev)
    batch = tuple(_t(x, dev) for x in (s, s2, a, rew, done, trunc, nst))
    ctx.dbg_set_sac_noise(_t(e1, dev), _t(e2, dev))
    try:
        ctx.prof_begin()
        key, cnt = ctx.fastsac_critic_update(pd, P, qd, Q, qm, qv, QT, lad, am, av, batch, _t(scale, dev), L.prng_key(5), 0, hp, met)
        ctx.prof_end()
        gemm = [q for q in ctx.prof_rows() if q["kernel"] in ("k_gemm_fwd", "k_gemm_dx", "k_gemm_dw")]
        assert gemm and not any(q["engine"] == 0 for q in gemm), [q for q in gemm if q["engine"] == 0]   # no silent exact-fp32 fallback at the bench batch
        m = met.cpu().numpy().astype(np.float64)
        assert m[0] == pytest.approx(r["q_loss"], rel=1e-5) and m[4] == pytest.approx(r["entropy"], rel=1e-5)
        gq_e = np.concatenate([r["g_q1"], r["g_q2"]])
        gq_d = qm.cpu().numpy().astype(np.float64) / (1.0 - h["adam_beta1"])
        rel = np.linalg.norm(gq_d - gq_e) / np.linalg.norm(gq_e)
        print(f"FastSAC critic step at B={B}: ||dg||/||g|| = {rel:.2e}, q_loss rel err {abs(m[0] - r['q_loss']) / abs(r['q_loss']):.1e}")
        assert rel < 1e-5
        n = qflat[0].size
        for k in range(2):
            for name, off, ln in ofs.blocks(O + A, ofs.CRITIC_HIDDEN, NA):
                ref = gq_e[k * n + off:k * n + off + ln]
                assert np.linalg.norm(gq_d[k * n + off:k * n + off + ln] - ref) <= 2e-5 * np.linalg.norm(ref) + 1e-12, (k, name)
        # policy step on the device's updated critics; the oracle gets exactly those parameters
        qa, la2 = Q.cpu().numpy().astype(np.float64), float(lad[0])
        p = ofs.policy_step(pflat.astype(np.float64), qa[:n], qa[n:], la2, O, A, NA, s.astype(np.float64), e2, scale, h, False)
        key, pcnt = ctx.fastsac_policy_update(pd, P, pm, pv, qd, Q, lad, batch[0], _t(scale, dev), key, 0, hp, pmet)
        pmv = pmet.cpu().numpy().astype(np.float64)
        assert pmv[0] == pytest.approx(p["policy_loss"], rel=1e-5, abs=1e-6)
        gp_d = pm.cpu().numpy().astype(np.float64) / (1.0 - h["adam_beta1"])
        relp = np.linalg.norm(gp_d - p["g_policy"]) / np.linalg.norm(p["g_policy"])
        print(f"FastSAC policy step at B={B}: ||dg||/||g|| = {relp:.2e}")
        assert relp < 1e-5
        assert pmv[2] == pytest.approx(np.linalg.norm(p["g_policy"]), rel=1e-5)
    finally:
        ctx.dbg_set_sac_noise(None, None)


def test_two_stream_schedule_is_bit_identical_to_one_stream(ctx, dev):
    """Option two_streams = 0 issues every pass on the caller's stream; the default runs critic 2 (and the online passes of the critic
    update) on the side stream.  Same kernels, same reduction order: parameters, moments and metrics have to agree bit for bit."""
    from rlx_amd.hip import lib as L
    rng = np.random.default_rng(7)
    O, A, NA, B = 48, 12, 101, 8192
    h = dict(gamma=0.97, tau=0.125, v_min=-20.0, v_max=20.0, log_std_min=-5.0, log_std_max=0.0, learning_rate=3e-4, weight_decay=0.001,
             adam_beta1=0.9, adam_beta2=0.95, target_entropy=0.0, log_alpha=float(np.log(0.05)))
    pflat, qflat = ofs.make_params(33, O, A, NA)
    scale = np.linspace(0.5, 1.5, A).astype(np.float32)
    f32 = lambda x: np.asarray(x, dtype=np.float32)
    s, s2 = f32(rng.standard_normal((B, O))), f32(rng.standard_normal((B, O)))
    a = f32(np.tanh(rng.standard_normal((B, A))) * scale)
    rew, done = f32(3.0 * rng.standard_normal(B)), f32(rng.random(B) < 0.2)
    trunc, nst = f32((rng.random(B) < 0.5) * done), f32(rng.integers(1, 4, B))
    pd, qd = lnmlp_desc(O, ofs.POLICY_HIDDEN, 2 * A), lnmlp_desc(O + A, ofs.CRITIC_HIDDEN, NA)
    hp = _hp(h, NA, True)
    batch = tuple(_t(x, dev) for x in (s, s2, a, rew, done, trunc, nst))

    def run(two):
        ctx.set_option("two_streams", two)
        P, Q, QT = _t(pflat, dev), _t(np.concatenate(qflat[:2]), dev), _t(np.concatenate(qflat[2:]), dev)
        zl = torch.zeros_like
        qm, qv, pm, pv = zl(Q), zl(Q), zl(P), zl(P)
        lad, am, av = _t([h["log_alpha"]], dev), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        met, pmet = torch.zeros(8, device=dev), torch.zeros(3, device=dev)
        key, cnt, pcnt = L.prng_key(11), 0, 0
        for _ in range(3):
            key, cnt = ctx.fastsac_critic_update(pd, P, qd, Q, qm, qv, QT, lad, am, av, batch, _t(scale, dev), key, cnt, hp, met)
            key, pcnt = ctx.fastsac_policy_update(pd, P, pm, pv, qd, Q, lad, batch[0], _t(scale, dev), key, pcnt, hp, pmet)
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in (P, Q, QT, qm, qv, pm, pv, lad, met, pmet)]
    try:
        one, two = run(0), run(1)
    finally:
        ctx.set_option("two_streams", 1)
    for x, y, name in zip(one, two, ("policy", "critics", "targets", "qm", "qv", "pm", "pv", "log_alpha", "critic metrics", "policy metrics")):
        assert np.isfinite(x).all() and np.array_equal(x, y), name


def test_nstep_replay_sample_matches_the_reference_ring(ctx, dev):
    """rlx_fastsac_replay_sample_f32 on the rings the reference's own ReplayBuffer produced (n = 1; n = 3 before and after the
    ring wrapped), with the indices its torch.randint calls returned."""
    z = np.load(FIX)
    names = ("states", "next_states", "actions", "rewards", "dones", "truncations")
    for tag in ("n1", "n3_partial", "n3_full"):
        ring = tuple(_t(z[tag + "_ring_" + k], dev) for k in names)
        it, ie = (torch.from_numpy(z[tag + k].astype(np.int32)).to(dev) for k in ("_idx_t", "_idx_e"))
        B, O, A = it.numel(), ring[0].shape[2], ring[2].shape[2]
        out = (torch.empty(B, O, device=dev), torch.empty(B, O, device=dev), torch.empty(B, A, device=dev)) + tuple(
            torch.empty(B, device=dev) for _ in range(4))
        ctx.fastsac_replay_sample(ring, int(z[tag + "_n_steps"]), float(z["ring_gamma"]), int(z[tag + "_pos"]), int(z[tag + "_size"]), it, ie, out)
        for name, got in zip(names + ("effective_n_steps",), out):
            np.testing.assert_allclose(got.cpu().numpy(), z[tag + "_" + name], rtol=1e-6, atol=1e-7, err_msg=tag + " " + name)


def _fastsac_plugin(env_over, alg_over, pidx=None, cidx=None):
    import rlx_amd.algorithms.fastsac.hip  # noqa: F401
    from test_gpu_obs_indices import _plugin
    return _plugin("fastsac.hip", env_over, alg_over, pidx, cidx)


@pytest.mark.parametrize("n_steps,indices", [(1, False), (3, False), (1, True)])
def test_plugin_trains_on_the_synthetic_env(dev, tmp_path, n_steps, indices):
    """A few vector steps of `fastsac.hip` end to end: acting, the ring, n-step sampling, normaliser, 2 x 2 critic / policy cadence,
    logging, evaluation, checkpoint round trip."""
    pidx, cidx = (np.arange(0, 10), np.arange(6, 24)) if indices else (None, None)
    cls, config, env = _fastsac_plugin(dict(nr_envs=32, obs_dim=24, act_dim=4, horizon=12),
                                       dict(batch_size=64, buffer_size_per_env=8, learning_starts=3, n_steps=n_steps, nr_atoms=51,
                                            nr_critic_updates_per_policy_update=2, nr_policy_updates_per_step=2,
                                            total_timesteps=32 * 12, logging_frequency=32 * 4, evaluation_frequency=32 * 8,
                                            save_frequency=32 * 4), pidx, cidx)
    config.runner.save_model = True
    m = cls(config, env, env, str(tmp_path), None)
    assert (m.pdesc.in_dim, m.qdesc.in_dim, m.qdesc.out_dim) == ((10, 18 + 4, 51) if indices else (24, 28, 51))
    p0, q0, t0 = m.pparams.clone(), m.qparams.clone(), m.qtarget.clone()
    m.train()
    assert all(np.isfinite(v) for v in m.last_metrics.values()), m.last_metrics
    assert m.critic_count == 9 * 4 and m.policy_count == 9 * 2          # steps 4..12 optimise: 2 x 2 critic, 2 policy updates each
    assert (m.pparams - p0).abs().max().item() > 0 and (m.qparams - q0).abs().max().item() > 0 and (m.qtarget - t0).abs().max().item() > 0
    assert m.size == 8 and m.pos == 12 % 8                               # the ring wrapped
    if m.obs_norm:
        assert int(m.norm_count[0]) == 9 * 2 * 4 * 64                    # every sampled state and next state counted once
    assert "eval/episode_return" in m.last_metrics
    # checkpoint round trip
    path = os.path.join(str(tmp_path), "models", "latest.model")
    assert os.path.exists(path)
    config.runner.load_model = path
    m2 = cls.load(config, env, env, str(tmp_path), None, [])
    m.save()
    m3 = cls.load(config, env, env, str(tmp_path), None, [])
    for k in cls._STATE:
        assert torch.equal(getattr(m3, k), getattr(m, k)), k
    assert m2.critic_count > 0 and m3.critic_count == m.critic_count
    assert len(m3.test(2)) <= 2


def test_runner_trains_and_tests_fastsac_from_the_command_line(monkeypatch, tmp_path):
    """`--algorithm.name=fastsac.hip` through the Runner (rl_x/runner/runner.py): train with save_model, then `test` mode from the
    checkpoint restores the exact parameters."""
    import sys
    from rlx_amd.runner.runner import Runner
    monkeypatch.chdir(tmp_path)
    base = ["experiment.py", "--algorithm.name=fastsac.hip", "--environment.name=synthetic.random_obs", "--environment.nr_envs=32",
            "--environment.obs_dim=20", "--environment.act_dim=3", "--environment.horizon=6"]
    flags = ["--algorithm.batch_size=64", "--algorithm.buffer_size_per_env=8", "--algorithm.learning_starts=2", "--algorithm.nr_atoms=31",
             "--algorithm.total_timesteps=320", "--algorithm.logging_frequency=64", "--algorithm.save_frequency=64"]
    monkeypatch.setattr(sys, "argv", base + ["--runner.mode=train", "--runner.save_model=true", "--runner.run_name=fsac"] + flags)
    trained = Runner().run()
    path = os.path.join(trained.save_path, "latest.model")
    assert os.path.exists(path) and trained.critic_count == 8 * 8 and trained.policy_count == 8 * 2
    monkeypatch.setattr(sys, "argv", base + ["--runner.mode=test", f"--runner.load_model={path}", "--runner.nr_test_episodes=2"])
    tested = Runner().run()
    ckpt = np.load(path, allow_pickle=False)
    assert torch.equal(tested.pparams.cpu(), torch.from_numpy(ckpt["pparams"])) and tested.critic_count == int(ckpt["critic_count"]) > 0
