"""GPU: the multi-GPU minibatch protocol through the HIP kernels, emulating 2 and 4 ranks on
one device: local rows + global statistics (phase 2) + summed contributions == the single
device result (gradients within 1e-5, identical permutations by construction)."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo
from oracle.sharding import local_minibatches
from rlx_amd.hip import PpoHparams, mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_contributions_sum_to_single_device(ctx, dev, world):
    rng = np.random.default_rng(world)
    T, NG, O, A, mb = 8, 64, 17, 6, 256
    ps, cs = nets.make_spec("B", O, A, True), nets.make_spec("B", O, 1, False)
    pd = mlp_desc(O, ps.hidden, A, ps.act, True, True)
    cd = mlp_desc(O, cs.hidden, 1, cs.act, True, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.05 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    states = rng.standard_normal((T, NG, O)).astype(np.float32)
    actions = rng.standard_normal((T, NG, A)).astype(np.float32)
    logp = (rng.standard_normal((T, NG)) * 0.1 - 8.5).astype(np.float32)
    returns = rng.standard_normal((T, NG)).astype(np.float32)
    adv = (rng.standard_normal((T, NG)) * 2 + 0.5).astype(np.float32)
    perm = rng.permutation(T * NG).astype(np.int32)
    idx = perm[:mb]
    hp = PpoHparams(0.1, 0.01, 0.7, 5.0, 0.9, 0.999, 1e-8)
    P, C = _t(pp, dev), _t(cp, dev)
    # single device
    pg1, cg1, met1 = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.zeros(8, device=dev)
    ctx.ppo_minibatch_fwd_bwd(pd, P, pg1, cd, C, cg1, met1, _t(states, dev), _t(actions, dev), _t(logp, dev),
                              _t(returns, dev), _t(adv, dev), _t(idx, dev), hp)
    # emulated ranks
    nl = NG // world
    stats = torch.zeros(4, dtype=torch.float64, device=dev)
    a = adv.reshape(-1)[idx].astype(np.float64)
    stats[0], stats[1], stats[2] = a.sum(), (a * a).sum(), mb       # == all-reduced phase-0 sums
    pg, cg, met = torch.zeros_like(pg1), torch.zeros_like(cg1), torch.zeros(8, device=dev)
    for r in range(world):
        sl = slice(r * nl, (r + 1) * nl)
        compact, counts, offsets = local_minibatches(torch.from_numpy(idx), 1, mb, NG, nl, r * nl)
        g_p, g_c, m = torch.zeros_like(pg1), torch.zeros_like(cg1), torch.zeros(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(pd, P, g_p, cd, C, g_c, m, _t(states[:, sl], dev), _t(actions[:, sl], dev),
                                  _t(logp[:, sl], dev), _t(returns[:, sl], dev), _t(adv[:, sl], dev),
                                  compact.to(dev), hp, mb_global=mb, stats_io=stats, phase=2)
        if r != 0:
            m[[2, 5, 6, 7]] = 0
        pg += g_p; cg += g_c; met += m            # what all_reduce(sum) does
    for got, exp in ((pg, pg1), (cg, cg1)):
        got, exp = got.cpu().numpy(), exp.cpu().numpy()
        assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 1e-5
    np.testing.assert_allclose(met.cpu().numpy(), met1.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_golden_minibatch_fixture(ctx, dev):
    """HIP vs the committed golden vectors (tests/golden/minibatch_*.npz)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for arch in "AB":
        m = np.load(os.path.join(gold, f"minibatch_{arch}.npz"))
        ps, cs = nets.make_spec(arch, 17, 6, True), nets.make_spec(arch, 17, 1, False)
        pd = mlp_desc(17, ps.hidden, 6, ps.act, ps.ln_first, True)
        cd = mlp_desc(17, cs.hidden, 1, cs.act, cs.ln_first, False)
        hp = PpoHparams(float(m["clip_range"]), float(m["entropy_coef"]), float(m["critic_coef"]), 0.5, 0.9, 0.999, 1e-8)
        P, C = _t(m["pparams"], dev), _t(m["cparams"], dev)
        pg, cg, met = torch.zeros_like(P), torch.zeros_like(C), torch.zeros(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, _t(m["states"], dev), _t(m["actions"], dev),
                                  _t(m["log_probs"], dev), _t(m["returns"], dev), _t(m["advantages"], dev),
                                  _t(m["idx"], dev), hp)
        mm = met.cpu().numpy()
        assert abs(mm[0] - float(m["pg_loss"])) <= 1e-5 * abs(float(m["pg_loss"])) + 1e-6
        assert abs(mm[1] - float(m["critic_loss"])) <= 1e-5 * abs(float(m["critic_loss"])) + 1e-6
        for got, exp in ((pg, m["pgrads"]), (cg, m["cgrads"])):
            assert np.linalg.norm(got.cpu().numpy() - exp) / np.linalg.norm(exp) < 1e-5
        # one optimizer step (clip 0.5 + Adam) against the golden post-update parameters
        pm, pv = torch.zeros_like(P), torch.zeros_like(P)
        nrm = torch.zeros(1, device=dev)
        ctx.clip_adam_step(P, pg, pm, pv, 1, float(m["lr"]), float(m["max_grad_norm"]), grad_norm_out=nrm)
        np.testing.assert_allclose(nrm.item(), float(m["policy_grad_norm"]), rtol=1e-5)
        d = np.abs(P.cpu().numpy() - m["pparams_after"])
        assert (d <= 2e-6).mean() > 0.999 and d.max() <= 2 * float(m["lr"]) + 1e-6
