"""GPU: the HIP path, through the C ABI, against golden vectors produced by EXECUTING the reference's own PyTorch flavour
(tests/golden/reference_{ppo,sac}_{f64r,f32}.npz, written by tests/golden/make_reference_golden.py -- networks
ppo/pytorch/policy.py + critic.py, closures of ppo/pytorch/ppo.py:98-166, sac/pytorch/policy.py).  No oracle in between.

THE PARITY BAR (1e-5 relative on losses / advantages / gradients) is asserted against the "f64r" fixtures: the reference run
in float64 on float32-representable inputs, i.e. exact-arithmetic results for bit-identical inputs -- the only error left is
the fp32 kernels' own.  The "f32" fixtures (the reference in fp32, which carries its own rounding of the same size as ours)
are kept as a second, looser witness (2e-5 .. 5e-5: two fp32 evaluations of one quantity differ by the sum of their errors).

The one place where the PyTorch flavour's arithmetic differs from the JAX flavour the kernels implement is the advantage
normalisation (`Tensor.std()` unbiased vs `jnp.std` population): the test hands the minibatch entry point the statistics
it would have all-reduced (`stats_io`, phase 2) with the second moment set so that the kernel's population formula
yields the unbiased value.  torch's gradient clip scales by c / (norm + 1e-6) instead of c / norm: a 1e-6 relative
difference, below the bar."""
import os

import numpy as np
import pytest
import torch

from rlx_amd.hip import PpoHparams, mlp_desc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_TANH, ACT_RELU = 0, 2


def _t(a, dev, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


# tolerances: (relative on losses / advantages / values, gradients as ||d|| / ||g||, approx-KL relative, SAC)
TOL = {"f64r": dict(loss=1e-5, grad=1e-5, kl=1e-5, sac=1e-5), "f32": dict(loss=2e-5, grad=2e-5, kl=1e-4, sac=5e-5)}


@pytest.mark.parametrize("tag", ["f64r", "f32"])
def test_ppo_reference_fixture(ctx, dev, tag):
    g = np.load(os.path.join(GOLDEN, "reference_ppo_%s.npz" % tag))
    tol = TOL[tag]
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    T, N = g["rewards"].shape
    pd = mlp_desc(O, [H, H], A, ACT_TANH, False, True)
    cd = mlp_desc(O, [H, H], 1, ACT_TANH, False, False)
    states, next_states = _t(g["states"], dev), _t(g["next_states"], dev)
    # networks (policy.py:44-50, critic.py:27-33)
    mean = torch.empty(T * N, A, device=dev)
    ctx.mlp_fwd(pd, _t(g["pparams0"], dev), states.view(-1, O), mean)
    np.testing.assert_allclose(mean.cpu().numpy().reshape(T, N, A), g["mean"], rtol=1e-5, atol=2e-6)
    values, next_values = torch.empty(T * N, 1, device=dev), torch.empty(T * N, 1, device=dev)
    C0 = _t(g["cparams0"], dev)
    ctx.mlp_fwd(cd, C0, states.view(-1, O), values)
    ctx.mlp_fwd(cd, C0, next_states.view(-1, O), next_values)
    np.testing.assert_allclose(values.cpu().numpy().reshape(T, N), g["values"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(next_values.cpu().numpy().reshape(T, N), g["next_values"], rtol=1e-5, atol=2e-6)
    # GAE (ppo.py:109-118) on the reference's own values
    adv, ret = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
    ctx.gae(_t(g["rewards"], dev), _t(g["values"], dev), _t(g["next_values"], dev), _t(g["terminations"], dev), adv, ret,
            float(g["gamma"]), float(g["gae_lambda"]))
    np.testing.assert_allclose(adv.cpu().numpy(), g["advantages_exact" if tag == "f64r" else "advantages"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), g["returns_exact" if tag == "f64r" else "returns"], rtol=1e-5, atol=2e-6)
    # two consecutive minibatch updates (ppo.py:121-166): losses, gradients, clip + Adam
    clip, ec, cc, mgn, lr = (float(g[k]) for k in ("clip_range", "entropy_coef", "critic_coef", "max_grad_norm", "learning_rate"))
    hp = PpoHparams(clip, ec, cc, mgn, 0.9, 0.999, 1e-8)
    P, C = _t(g["pparams1"], dev), _t(g["cparams0"], dev)
    pm, pv, cm, cv = (torch.zeros_like(x) for x in (P, P, C, C))
    actions, logp = _t(g["actions"], dev), _t(g["log_probs"], dev)
    advantages, returns = _t(g["advantages"], dev), _t(g["returns"], dev)
    for step in range(2):
        s = "_%d" % step
        idx = g["idx" + s]
        a = g["advantages"].reshape(-1)[idx].astype(np.float64)
        n = a.size
        m_, var_unbiased = a.mean(), a.var(ddof=1)
        stats = torch.tensor([a.sum(), n * (var_unbiased + m_ * m_), float(n), 0.0], dtype=torch.float64, device=dev)
        pg, cg, met = torch.empty_like(P), torch.empty_like(C), torch.empty(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, returns, advantages, _t(idx, dev, np.int32), hp,
                                  mb_global=n, stats_io=stats, phase=2)
        m = met.cpu().numpy()
        # step 1 starts from the parameters of OUR step-0 Adam update (fp32), the fixture's from the reference's: the second
        # step's inputs differ by that rounding (~1e-8 absolute per parameter), far below the bar
        np.testing.assert_allclose(m[0], g["pg_loss" + s], rtol=tol["loss"], atol=2e-7)
        np.testing.assert_allclose(cc * m[1], g["critic_loss" + s], rtol=tol["loss"], atol=2e-7)
        np.testing.assert_allclose(m[2], g["entropy_loss" + s], rtol=1e-5, atol=1e-6)
        # approx KL = mean((ratio - 1) - log ratio): a difference of O(1) fp32 numbers (resolution 6e-8 each) averaged over 64
        np.testing.assert_allclose(m[3], g["approx_kl" + s], rtol=tol["kl"], atol=2e-8)
        np.testing.assert_allclose(m[4], g["clip_fraction" + s], atol=1.5 / n)
        for grads, name, norm in ((pg, "pgrads_clipped", float(g["policy_grad_norm" + s])), (cg, "cgrads_clipped", float(g["critic_grad_norm" + s]))):
            exp = g[name + s].astype(np.float64)
            got = grads.cpu().numpy().astype(np.float64) * min(1.0, mgn / (norm + 1e-6))
            assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < tol["grad"]
        gn = torch.empty(2, device=dev)
        ctx.clip_adam_step(P, pg, pm, pv, step + 1, lr, mgn, grad_norm_out=gn[0:1])
        ctx.clip_adam_step(C, cg, cm, cv, step + 1, lr, mgn, grad_norm_out=gn[1:2])
        np.testing.assert_allclose(gn.cpu().numpy(), [float(g["policy_grad_norm" + s]), float(g["critic_grad_norm" + s])], rtol=tol["grad"])
        for got, name in ((P, "pparams_after"), (C, "cparams_after")):
            d = np.abs(got.cpu().numpy() - g[name + s])
            assert d.max() <= 2 * lr * (step + 1) and (d < 2e-6).mean() > 0.99, (d.max(), (d < 2e-6).mean())


def test_sac_policy_reference_fixture(ctx, dev):
    """sac/pytorch/policy.py:66-73 (deterministic action) through rlx_sac_act_f32; the stochastic entry points draw their
    noise from the threefry stream, so the noise-dependent quantities are compared through the oracle, itself pinned on
    this fixture (tests/test_oracle_reference_pin.py)."""
    g = np.load(os.path.join(GOLDEN, "reference_sac_f64r.npz"))
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    B = g["next_states"].shape[0]
    d = mlp_desc(O, [H, H], 2 * A, ACT_RELU, False, False)
    act = torch.empty(B, A, device=dev)
    ctx.sac_act(d, _t(g["pparams"], dev), _t(g["next_states"], dev), np.array([1, 2], np.uint32), act, float(g["log_std_min"]),
                float(g["log_std_max"]), deterministic=True)
    np.testing.assert_allclose(act.cpu().numpy(), g["deterministic_action"], rtol=1e-5, atol=2e-6)
    # the critic pair (q_network.py:22-41) through the generic MLP entry
    qd = mlp_desc(O + A, [H, H], 1, ACT_RELU, False, False)
    x = torch.cat([_t(g["states"], dev), _t(g["actions"], dev)], dim=1).contiguous()
    n = g["qparams"].size // 2
    for k, name in enumerate(("q1", "q2")):
        q = torch.empty(B, 1, device=dev)
        ctx.mlp_fwd(qd, _t(g["qparams"][k * n:(k + 1) * n], dev), x, q)
        np.testing.assert_allclose(q.cpu().numpy().reshape(-1), g[name], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("tag", ["f64r", "f32"])
def test_sac_update_reference_fixture(ctx, dev, tag):
    """rlx_sac_update_f32 against the reference's SAC closures executed (sac/pytorch/sac.py:90-166 at the fixture's
    parameters), fed the noise that run consumed (rlx_dbg_set_sac_noise): the three losses, entropy, alpha, the
    Q value, all gradients (= 10 x the first Adam moments) and log_alpha after its Adam step."""
    from rlx_amd.hip import SacHparams
    g = np.load(os.path.join(GOLDEN, "reference_sac_%s.npz" % tag))
    rt = TOL[tag]["sac"]
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    pd = mlp_desc(O, [H, H], 2 * A, ACT_RELU, False, False)
    qd = mlp_desc(O + A, [H, H], 1, ACT_RELU, False, False)
    P, Q, QT = _t(g["pparams"], dev), _t(g["qparams"], dev), _t(g["qtarget"], dev)
    LA = _t(np.array([g["log_alpha"]], np.float32), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    lr = float(g["learning_rate"])
    hp = SacHparams(float(g["gamma"]), float(g["tau"]), float(g["target_entropy"]), float(g["log_std_min"]),
                    float(g["log_std_max"]), lr, lr, lr, 0.9, 0.999, 1e-8)
    met = torch.zeros(10, device=dev)
    batch = tuple(_t(g[k], dev) for k in ("states", "next_states", "actions", "rewards", "terminations"))
    e_next, e_cur = _t(g["noise_next"], dev), _t(g["noise_cur"], dev)
    try:
        ctx.dbg_set_sac_noise(e_next, e_cur)
        ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, np.array([3, 4], np.uint32), 0, hp, met, 1)
        torch.cuda.synchronize()
    finally:
        ctx.dbg_set_sac_noise(None, None)
    m = met.cpu().numpy()
    for i, n in enumerate(("q_loss", "policy_loss", "entropy_loss", "entropy", "alpha", "min_q_mean")):
        assert m[i] == pytest.approx(float(np.asarray(g[n]).reshape(-1)[0]), rel=rt, abs=rt), n
    assert np.linalg.norm(pm.cpu().numpy() * 10 - g["gpolicy"]) / np.linalg.norm(g["gpolicy"]) < rt
    assert np.linalg.norm(qm.cpu().numpy() * 10 - g["gcritic"]) / np.linalg.norm(g["gcritic"]) < rt
    assert am.item() * 10 == pytest.approx(float(g["g_log_alpha"]), rel=10 * rt)
    assert LA.item() == pytest.approx(float(g["log_alpha_after"]), abs=1e-6)
    d = np.abs(Q.cpu().numpy() - g["qparams_after"])
    assert d.max() <= 2 * lr + 1e-6 and (d < 2e-6).mean() > 0.99


def test_replay_ring_reference_fixture(ctx, dev):
    """The HBM replay ring + rlx_sac_replay_sample_f32 against the JAX flavour's own numpy ReplayBuffer
    (sac/flax/replay_buffer.py, executed by make_reference_golden.py): same adds, same PCG64 draws, identical rows."""
    g = np.load(os.path.join(GOLDEN, "reference_sac_replay.npz"))
    NE, B = int(g["nr_envs"]), int(g["batch"])
    steps, _, O = g["add_states"].shape
    A = g["add_actions"].shape[2]
    cap = int(g["capacity"]) // NE                                            # replay_buffer.py:8
    names = ("states", "next_states", "actions", "rewards", "terminations")
    ring = (torch.zeros(cap, NE, O, device=dev), torch.zeros(cap, NE, O, device=dev), torch.zeros(cap, NE, A, device=dev),
            torch.zeros(cap, NE, device=dev), torch.zeros(cap, NE, device=dev))
    rng = np.random.default_rng(int(g["sampler_seed"]))
    pos = size = 0
    for t in range(steps):
        for buf, k in zip(ring, names):
            buf[pos] = _t(g["add_" + k][t], dev)
        pos, size = (pos + 1) % cap, min(size + 1, cap)
        if ("sample%d_states" % t) in g.files:
            i1 = rng.integers(size, size=B)                                   # replay_buffer.py:31-32
            i2 = rng.integers(NE, size=B)
            out = (torch.empty(B, O, device=dev), torch.empty(B, O, device=dev), torch.empty(B, A, device=dev),
                   torch.empty(B, device=dev), torch.empty(B, device=dev))
            ctx.sac_replay_sample(ring, _t(i1, dev, np.int32), _t(i2, dev, np.int32), out)
            for k, x in zip(names, out):
                assert np.array_equal(x.cpu().numpy(), g["sample%d_%s" % (t, k)]), (t, k)
    assert pos == int(g["final_pos"]) and size == int(g["final_size"])
