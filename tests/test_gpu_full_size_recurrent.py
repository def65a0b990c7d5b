"""GPU: BASELINE.json configs[4] at FULL size (PPO+LSTM / PPO+GRU, 2048 envs x 128 steps, obs 17 / act 6,
minibatch 32768 = 256 envs x 128 steps, 10 epochs -> 80 updates; rl_x/algorithms/ppo_lstm/flax_full_jit/ppo_lstm.py:181-263).
The float64 autograd oracle cannot evaluate a 256-env x 128-step minibatch in seconds, so the full shape is tied to it in two
steps:
  * shard additivity over env subsets: the 256-env minibatch is laid out as 32 groups of 8 envs that share one advantage
    block (so every group -- and the whole minibatch -- normalises its advantages with the same mean / std); the loss is a
    mean over envs, hence grad(256 envs) == mean over the 32 groups of grad(group), metrics likewise;
  * two of those 8-env x 128-step groups are compared with the oracle directly (full sequence length: the BPTT depth is
    what the small tests in test_gpu_ppo_lstm.py do not reach).
Plus size-independent properties of the full shape: bit-for-bit determinism of the minibatch kernels and of the whole
80-update call, the carry reset at `done` (outputs after a reset do not depend on the initial carry), and the env-index
permutation rows being bijections of [0, 2048) reproduced bit-exactly from the key chain."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo_lstm as ol, prng
from rlx_amd.hip import PpoHparams, mlp_desc
from rlx_amd.hip.lib import lstm_policy_desc

pytestmark = pytest.mark.gpu

T, N, O, A, MBS, E = 128, 2048, 17, 6, 32768, 10
NE = MBS // T          # 256 envs per minibatch
GROUP = 8


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _hp():
    hp = PpoHparams()
    hp.clip_range, hp.entropy_coef, hp.critic_coef, hp.max_grad_norm = 0.2, 0.01, 0.5, 0.5
    hp.adam_b1, hp.adam_b2, hp.adam_eps = 0.9, 0.999, 1e-5
    return hp


def _setup(cell, seed=0):
    rng = np.random.default_rng(seed)
    spec = ol.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), False, cell)
    p = ol.init_params(spec, rng, 1.0)
    p = (p + 0.03 * rng.standard_normal(p.shape)).astype(np.float32)
    cs = nets.make_spec("B", O, 1, False)
    cp = (nets.init_params(cs, rng, 1.0) + 0.03 * rng.standard_normal(cs.n_params)).astype(np.float32)
    ld = lstm_policy_desc(O, A, 128, 64, (512, 256, 128), False, 1 if cell == "gru" else 0)
    cd = mlp_desc(O, cs.hidden, 1, cs.act, True, False)
    return rng, spec, p, cs, cp, ld, cd


def _case(rng, spec, p, n_envs, p_done=0.01):
    """Rollout arrays [T, n_envs, .]; advantages repeat with period GROUP along the env axis (see the module docstring).
    log_probs are the policy's own log-probs plus noise, so ratios straddle the clip range."""
    states = rng.standard_normal((T, n_envs, O)).astype(np.float32)
    actions = rng.standard_normal((T, n_envs, A)).astype(np.float32)
    dones = (rng.random((T, n_envs)) < p_done).astype(np.float32)
    c0 = (0.5 * rng.standard_normal((n_envs, spec.H))).astype(np.float32)
    h0 = np.tanh(0.5 * rng.standard_normal((n_envs, spec.H))).astype(np.float32)
    returns = rng.standard_normal((T, n_envs)).astype(np.float32)
    base = (rng.standard_normal((T, GROUP)) * 2 + 0.3).astype(np.float32)
    adv = np.tile(base, (1, n_envs // GROUP))
    return states, actions, returns, adv, dones, c0, h0


def _own_logp(spec, p, case, cols, rng):
    """float64 oracle log-probs of the stored actions for the env columns `cols` (+ noise): a forward pass only."""
    states, actions = case[0][:, cols], case[1][:, cols]
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    with torch.no_grad():
        mean = ol.forward_sequence(spec, t64(p), t64(states), t64(case[4][:, cols]), t64(case[5][cols]), t64(case[6][cols])).numpy()
    logstd = p[spec.off["logstd"][0]:][:A].astype(np.float64)[None, None, :]
    lp = (-0.5 * ((actions - mean) / np.exp(logstd)) ** 2 - 0.5 * ol.LOG_2PI - logstd).sum(-1)
    return (lp + 0.05 * rng.standard_normal(lp.shape)).astype(np.float32)


def _run_mb(ctx, dev, ld, cd, spec, cs, P, C, dcase, env_idx, hp):
    pg, cg, met = torch.empty(spec.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
    ctx.ppo_lstm_minibatch_fwd_bwd(ld, P, pg, cd, C, cg, met, *dcase, _t(np.asarray(env_idx, dtype=np.int32), dev), hp)
    return pg, cg, met


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_full_size_minibatch_additive_deterministic_and_oracle_pinned(ctx, dev, cell):
    rng, spec, p, cs, cp, ld, cd = _setup(cell)
    states, actions, returns, adv, dones, c0, h0 = case = _case(rng, spec, p, N)
    # the minibatch: 256 distinct envs, group g = envs [8 g', 8 g' + 8) of the rollout so that the advantage block repeats
    groups = rng.permutation(N // GROUP)[: NE // GROUP]
    env_idx = (groups[:, None] * GROUP + np.arange(GROUP)[None, :]).reshape(-1)
    logp = np.zeros((T, N), dtype=np.float32)
    logp[:, env_idx] = _own_logp(spec, p, case, env_idx, rng)
    hp = _hp()
    dcase = [_t(x, dev) for x in (states, actions, logp, returns, adv, dones, c0, h0)]
    P, C = _t(p, dev), _t(cp, dev)
    pg, cg, met = _run_mb(ctx, dev, ld, cd, spec, cs, P, C, dcase, env_idx, hp)
    pg2, cg2, met2 = _run_mb(ctx, dev, ld, cd, spec, cs, P, C, dcase, env_idx, hp)
    assert torch.equal(pg, pg2) and torch.equal(cg, cg2) and torch.equal(met, met2)      # fixed-order reductions
    assert bool(torch.isfinite(pg).all()) and bool(torch.isfinite(cg).all())
    # ---- shard additivity: mean over the 32 groups (fp64 accumulation of fp32 results)
    spg = torch.zeros(spec.n_params, device=dev, dtype=torch.float64)
    scg = torch.zeros(cs.n_params, device=dev, dtype=torch.float64)
    smet = torch.zeros(8, device=dev, dtype=torch.float64)
    group_out = []
    for g in range(NE // GROUP):
        gp, gc, gm = _run_mb(ctx, dev, ld, cd, spec, cs, P, C, dcase, env_idx[g * GROUP:(g + 1) * GROUP], hp)
        spg += gp.double(); scg += gc.double(); smet += gm.double()
        if g < 2:
            group_out.append((gp.cpu().numpy().astype(np.float64), gc.cpu().numpy().astype(np.float64), gm.cpu().numpy()))
    ng = NE // GROUP
    spg /= ng; scg /= ng; smet /= ng
    for name, (o, n) in spec.off.items():                                              # per parameter block
        ref = spg[o:o + n]
        err = (pg[o:o + n].double() - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1e-6) + 1e-8, (name, err)
    assert (cg.double() - scg).abs().max().item() <= 2e-5 * scg.abs().max().item()
    m, sm = met.cpu().numpy(), smet.cpu().numpy()
    np.testing.assert_allclose(m[[0, 1, 3, 4]], sm[[0, 1, 3, 4]], rtol=1e-5, atol=1e-7)   # pg, critic, kl, clip fraction
    np.testing.assert_allclose(m[[2, 5, 6, 7]], sm[[2, 5, 6, 7]], rtol=1e-6, atol=1e-7)   # replicated values
    # ---- oracle on two of the 8-env x 128-step groups (float64 autograd through all 128 steps)
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    for g in range(2):
        cols = env_idx[g * GROUP:(g + 1) * GROUP]
        Pt, Ct = t64(p).requires_grad_(True), t64(cp).requires_grad_(True)
        a = adv[:, cols].astype(np.float64)
        a = (a - a.mean()) / (a.std() + 1e-8)
        loss, mo = ol.ppo_lstm_loss(spec, Pt, cs, Ct, t64(states[:, cols]), t64(actions[:, cols]), t64(logp[:, cols]),
                                    t64(returns[:, cols]), t64(a), t64(dones[:, cols]), t64(c0[cols]), t64(h0[cols]),
                                    hp.clip_range, hp.entropy_coef, hp.critic_coef)
        loss.backward()
        gp, gc, gm = group_out[g]
        assert gm[0] == pytest.approx(mo["loss/policy_gradient_loss"].item(), rel=1e-5, abs=1e-6)
        assert gm[1] == pytest.approx(mo["loss/critic_loss"].item(), rel=1e-5)
        assert gm[3] == pytest.approx(mo["policy_ratio/approx_kl"].item(), rel=1e-4, abs=1e-7)
        assert gm[4] == pytest.approx(mo["policy_ratio/clip_fraction"].item(), abs=1.5 / (T * GROUP))
        ref_p, ref_c = Pt.grad.numpy(), Ct.grad.numpy()
        for name, (o, n) in spec.off.items():
            scale = max(np.abs(ref_p[o:o + n]).max(), 1e-6)
            err = np.abs(gp[o:o + n] - ref_p[o:o + n]).max()
            assert err <= 5e-5 * scale + 1e-8, (cell, name, err, scale)     # worst single entry of the block (max norm)
            assert np.linalg.norm(gp[o:o + n] - ref_p[o:o + n]) <= 1e-5 * np.linalg.norm(ref_p[o:o + n]) + 1e-10, (cell, name, "L2")
        assert np.linalg.norm(gp - ref_p) / np.linalg.norm(ref_p) < 1e-5      # the 1e-5 bar on the whole gradient, L2-relative
        assert np.abs(gc - ref_c).max() <= 2e-5 * np.abs(ref_c).max()
        assert np.linalg.norm(gc - ref_c) / np.linalg.norm(ref_c) < 1e-5
        print(f"full-size recurrent group {g} ({cell}): ||dg||/||g|| policy {np.linalg.norm(gp - ref_p) / np.linalg.norm(ref_p):.2e} "
              f"critic {np.linalg.norm(gc - ref_c) / np.linalg.norm(ref_c):.2e}")


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_full_size_carry_reset_at_done(ctx, dev, cell):
    """forward_sequence (policy.py:134-142) multiplies the carry by (1 - done[t-1]) before step t.  With done[63] = 1 for
    every env of the minibatch, steps >= 64 cannot see anything that happened before step 64.  The loss is a mean over
    (t, env), so with advantages whose second half repeats the first (same normalisation statistics for T = 64 and
    T = 128) the contribution of steps >= 64 is  late(R) = 2 * full_T128(R) - half_T64(R)  for every metric and
    gradient.  Two rollouts that agree on steps >= 64 but differ before must give the same late(R); without the reset
    they must not."""
    rng, spec, p, cs, cp, ld, cd = _setup(cell, seed=3)
    states, actions, returns, adv, dones, c0, h0 = _case(rng, spec, p, N, p_done=0.0)
    adv[T // 2:] = adv[:T // 2]
    env_idx = rng.permutation(N)[:NE].astype(np.int32)
    logp = (-8.5 + 0.1 * rng.standard_normal((T, N))).astype(np.float32)
    hp = _hp()
    P, C = _t(p, dev), _t(cp, dev)
    states2 = states.copy()
    states2[: T // 2] = rng.standard_normal((T // 2, N, O)).astype(np.float32)       # different history before step 64
    c0b, h0b = (c0 + 1.0).astype(np.float32), (-h0).astype(np.float32)

    def late(st, c0v, h0v, dn):
        d = [_t(x, dev) for x in (st, actions, logp, returns, adv, dn)]
        cc, hh = _t(c0v, dev), _t(h0v, dev)
        full = _run_mb(ctx, dev, ld, cd, spec, cs, P, C, d + [cc, hh], env_idx, hp)
        half = _run_mb(ctx, dev, ld, cd, spec, cs, P, C, [x[: T // 2] for x in d] + [cc, hh], env_idx, hp)
        return [2.0 * f.double() - h.double() for f, h in zip(full, half)], full
    dn_reset = dones.copy()
    dn_reset[T // 2 - 1, :] = 1.0
    l1, f1 = late(states, c0, h0, dn_reset)
    l2, f2 = late(states2, c0b, h0b, dn_reset)
    assert not torch.equal(f1[0], f2[0])                         # the two rollouts do differ as a whole
    # policy gradient of the late half: identical up to fp32 summation noise of the two 32768-row sums it is built from
    scale = l1[0].abs().max().item()
    assert (l1[0] - l2[0]).abs().max().item() <= 3e-5 * scale, ((l1[0] - l2[0]).abs().max().item(), scale)
    for k in (0, 3, 4):                                          # pg loss, approx KL, clip fraction of the late half
        assert l1[2][k].item() == pytest.approx(l2[2][k].item(), rel=2e-5, abs=2e-6)
    # negative control: without the reset the history (and the stored carry) reaches the late half
    n1, _ = late(states, c0, h0, dones)
    n2, _ = late(states2, c0b, h0b, dones)
    assert (n1[0] - n2[0]).abs().max().item() > 100 * 3e-5 * scale     # two orders of magnitude above the reset case's bound


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_full_size_update_is_deterministic_and_permutation_is_exact(ctx, dev, cell):
    rng, spec, p, cs, cp, ld, cd = _setup(cell, seed=7)
    states, actions, returns, adv, dones, c0, h0 = _case(rng, spec, p, N)
    logp = (-8.5 + 0.1 * rng.standard_normal((T, N))).astype(np.float32)
    hp = _hp()
    hp.max_grad_norm = 5.0
    M = N // NE
    lr = np.full(E * M, 3e-4, dtype=np.float32)
    dcase = [_t(x, dev) for x in (states, actions, logp, returns, adv, dones, c0, h0)]
    key = prng.prng_key(21)

    def run():
        P, C = _t(p, dev), _t(cp, dev)
        z = lambda x: torch.zeros_like(x)
        met = torch.empty(E * M, 10, device=dev)
        k2, cnt = ctx.ppo_lstm_update(ld, P, z(P), z(P), cd, C, z(C), z(C), *dcase, E, MBS, key, 0, lr, hp, met)
        torch.cuda.synchronize()
        return P, C, met, k2, cnt
    a, b = run(), run()
    assert a[4] == E * M == 80
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(a[2]).all()) and bool(torch.isfinite(a[0]).all())
    assert (a[0] - _t(p, dev)).abs().max().item() > 1e-4
    # env-index permutation [E, N] of the call: bit-exact vs the oracle's restatement of jax.random.permutation
    k_exp, idx = ol.env_minibatch_indices(key, N, E, M, NE)
    assert np.array_equal(a[3], k_exp)
    perm = torch.empty(E * N, dtype=torch.int32, device=dev)
    k3 = ctx.permutation(key, perm, E, N)
    assert np.array_equal(k3, k_exp)
    assert np.array_equal(perm.cpu().numpy().reshape(E * M, NE), idx)
    rows = np.sort(perm.cpu().numpy().reshape(E, N), axis=1)
    assert (rows == np.arange(N)[None, :]).all()
