"""CPU, world_size 2 over gloo: the data-parallel PPO update protocol --
index sharding of the GLOBAL permutation, batched global advantage statistics, local
contributions scaled by 1/mb_global, ONE all-reduce(sum) per update -- reproduces the
single-process result.  The per-rank math here is the oracle's (no GPU in this container);
tests/test_gpu_sharded_update.py checks the same protocol through the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nets, ppo as oppo, prng
from oracle.sharding import local_minibatches

T, NG, O, A, E, MB = 6, 16, 5, 3, 2, 24
CLIP, ENT, CC = 0.1, 0.01, 0.7


def _global_problem():
    rng = np.random.default_rng(0)
    ps, cs = nets.make_spec("A", O, A, True, 64), nets.make_spec("A", O, 1, False, 64)
    pp = nets.init_params(ps, rng, 0.01, dtype=np.float64) + 0.05 * rng.standard_normal(ps.n_params)
    cp = nets.init_params(cs, rng, 1.0, dtype=np.float64) + 0.05 * rng.standard_normal(cs.n_params)
    states = rng.standard_normal((T, NG, O))
    actions = rng.standard_normal((T, NG, A))
    logp = rng.standard_normal((T, NG)) * 0.1 - 4.0
    returns = rng.standard_normal((T, NG))
    adv = rng.standard_normal((T, NG)) * 2 + 0.5
    _, idx = prng.ppo_minibatch_indices(prng.prng_key(1), T * NG, E, (T * NG) // MB, MB, True)
    return ps, pp, cs, cp, states, actions, logp, returns, adv, idx


def _reference_update(u):
    ps, pp, cs, cp, states, actions, logp, returns, adv, idx = _global_problem()
    i = idx[u]
    f = lambda a, d: a.reshape(-1, d)[i] if d else a.reshape(-1)[i]
    madv = oppo.normalize_advantages(f(adv, 0))
    return oppo.ppo_loss_and_grads(ps, pp, cs, cp, f(states, O), f(actions, A), f(logp, 0), f(returns, 0), madv,
                                   CLIP, ENT, CC)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ps, pp, cs, cp, states, actions, logp, returns, adv, idx = _global_problem()
        nl = NG // world
        off = rank * nl
        sl = slice(off, off + nl)
        ls, la, llp, lret, ladv = states[:, sl], actions[:, sl], logp[:, sl], returns[:, sl], adv[:, sl]   # this rank's shard
        n_mb = idx.shape[0]
        compact, counts, offsets = local_minibatches(torch.from_numpy(idx.reshape(-1).astype(np.int32)), n_mb, MB, NG, nl, off)
        compact = compact.numpy()
        # batched global statistics: one all-reduce for all minibatches
        stats = torch.zeros(n_mb, 3, dtype=torch.float64)
        for u in range(n_mb):
            a = ladv.reshape(-1)[compact[offsets[u]:offsets[u + 1]]]
            stats[u] = torch.tensor([a.sum(), (a * a).sum(), len(a)])
        dist.all_reduce(stats)
        assert torch.all(stats[:, 2] == MB)
        out = []
        for u in (0, n_mb - 1):
            li = compact[offsets[u]:offsets[u + 1]]
            mean = stats[u, 0].item() / MB
            std = np.sqrt(max(stats[u, 1].item() / MB - mean * mean, 0.0))
            madv = (ladv.reshape(-1)[li] - mean) / (std + 1e-8)
            _, met, gp, gc = oppo.ppo_loss_and_grads(ps, pp, cs, cp, ls.reshape(-1, O)[li], la.reshape(-1, A)[li],
                                                     llp.reshape(-1)[li], lret.reshape(-1)[li], madv, CLIP, ENT, CC)
            share = len(li) / MB                        # oracle means over local rows; contribution = mean * share
            flat = torch.from_numpy(np.concatenate([gp * share, gc * share,
                                                    [met["loss/policy_gradient_loss"] * share,
                                                     met["loss/critic_loss"] * share]]))
            dist.all_reduce(flat)                        # ONE collective per update
            out.append(flat.numpy())
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_update_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_mb = E * (T * NG) // MB
    for flat, u in zip(out, (0, n_mb - 1)):
        loss, met, gp, gc = _reference_update(u)
        exp = np.concatenate([gp, gc, [met["loss/policy_gradient_loss"], met["loss/critic_loss"]]])
        np.testing.assert_allclose(flat, exp, rtol=1e-9, atol=1e-12)


def test_local_minibatches_partition():
    perm = torch.from_numpy(np.random.default_rng(0).permutation(T * NG).astype(np.int32))
    seen = []
    for rank in range(4):
        nl = NG // 4
        compact, counts, offsets = local_minibatches(perm, (T * NG) // MB, MB, NG, nl, rank * nl)
        assert int(counts.sum()) == T * nl and offsets[-1] == T * nl
        loc = compact.numpy()
        glob = (loc // nl) * NG + (loc % nl) + rank * nl          # back to global indices
        seen.append(glob)
    assert sorted(np.concatenate(seen).tolist()) == list(range(T * NG))


# ---- SAC: rows of the global batch sharded over ranks, ONE all-reduce of [policy grads | critic grads | loss sums] per update
def _sac_problem():
    from oracle import sac as osac
    rng = np.random.default_rng(3)
    Os, As, Bg = 7, 3, 48
    ps, qs = osac.make_specs(Os, As, 32)
    pp = osac.lecun_normal_init(ps, rng).astype(np.float64) + 0.02 * rng.standard_normal(ps.n_params)
    qp = np.concatenate([osac.lecun_normal_init(qs, rng) for _ in range(2)]).astype(np.float64) + 0.02 * rng.standard_normal(2 * qs.n_params)
    qtp = qp + 0.01 * rng.standard_normal(qp.shape)
    s, s2 = rng.standard_normal((Bg, Os)), rng.standard_normal((Bg, Os))
    a, r, term = np.tanh(rng.standard_normal((Bg, As))), rng.standard_normal(Bg), (rng.random(Bg) < 0.2).astype(np.float64)
    _, e1, e2 = osac.sample_noise(prng.prng_key(4), Bg, As, True)          # per-sample keys of the GLOBAL batch: rank r uses its rows
    return ps, pp, qs, qp, qtp, s, s2, a, r, term, e1.astype(np.float64), e2.astype(np.float64), As, Bg


def _sac_worker(rank, world, port, q):
    from oracle import sac as osac
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ps, pp, qs, qp, qtp, s, s2, a, r, term, e1, e2, As, Bg = _sac_problem()
        sl = slice(rank * Bg // world, (rank + 1) * Bg // world)
        sums, gp, gq, _ = osac.loss_and_grads(ps, pp, qs, qp, qtp, np.float64(-0.3), s[sl], s2[sl], a[sl], r[sl], term[sl], e1[sl], e2[sl],
                                              0.99, -float(As), batch_global=Bg)
        flat = torch.from_numpy(np.concatenate([gp, gq, [sums["sum/q_loss"], sums["sum/min_q"], sums["sum/logp"]]]))
        dist.all_reduce(flat)                            # ONE collective per update
        if rank == 0:
            q.put(flat.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_sac_update_matches_single_process():
    from oracle import sac as osac
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sac_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ps, pp, qs, qp, qtp, s, s2, a, r, term, e1, e2, As, Bg = _sac_problem()
    met, gp, gq, ga = osac.loss_and_grads(ps, pp, qs, qp, qtp, np.float64(-0.3), s, s2, a, r, term, e1, e2, 0.99, -float(As))
    n = ps.n_params + 2 * qs.n_params
    np.testing.assert_allclose(flat[:n], np.concatenate([gp, gq]), rtol=1e-10, atol=1e-13)
    ql, mq, lp = flat[n:] / Bg
    alpha = np.exp(-0.3)
    assert ql == pytest.approx(met["loss/q_loss"], rel=1e-12) and mq == pytest.approx(met["q_value/q_value"], rel=1e-12)
    assert -lp == pytest.approx(met["entropy/entropy"], rel=1e-12)
    assert alpha * (-lp + As) == pytest.approx(float(ga), rel=1e-12)      # the entropy coefficient's gradient from the reduced sum


# ---- PPO+LSTM: env columns sharded, every rank takes its share of each sequence minibatch (rlx_ppo_lstm_update_f32 on a context
# with a communicator): global advantage statistics from one all-reduce, loss scaled by 1 / (T * ne_global), one gradient all-reduce
def _lstm_problem():
    from oracle import ppo_lstm as ol
    rng = np.random.default_rng(8)
    Tl, NGl, Ol, Al = 5, 8, 6, 2
    spec = ol.LstmPolicySpec(Ol, Al, 128, 64, (512, 256, 128), False, "lstm", "concat")
    p = ol.init_params(spec, rng, 1.0).astype(np.float64) + 0.02 * rng.standard_normal(spec.n_params)
    cs = nets.make_spec("B", Ol, 1, False)
    cp = nets.init_params(cs, rng, 1.0, dtype=np.float64) + 0.02 * rng.standard_normal(cs.n_params)
    obs, act = rng.standard_normal((Tl, NGl, Ol)), rng.standard_normal((Tl, NGl, Al))
    logp, ret, adv = rng.standard_normal((Tl, NGl)) * 0.1 - 3, rng.standard_normal((Tl, NGl)), rng.standard_normal((Tl, NGl)) * 2 + 1
    done = (rng.random((Tl, NGl)) < 0.2).astype(np.float64)
    c0, h0 = rng.standard_normal((NGl, 64)) * 0.3, rng.standard_normal((NGl, 64)) * 0.3
    return spec, p, cs, cp, obs, act, logp, ret, adv, done, c0, h0


def _lstm_grads(spec, p, cs, cp, obs, act, logp, ret, advn, done, c0, h0, scale):
    """gradients of scale * SUM over (t, env) of the per-sample loss terms (the oracle's loss is their mean)"""
    from oracle import ppo_lstm as ol
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    P, C = tt(p).requires_grad_(True), tt(cp).requires_grad_(True)
    loss, met = ol.ppo_lstm_loss(spec, P, cs, C, tt(obs), tt(act), tt(logp), tt(ret), tt(advn), tt(done), tt(c0), tt(h0), 0.2, 0.0, 0.5)
    (loss * obs.shape[0] * obs.shape[1] * scale).backward()
    return np.concatenate([P.grad.numpy(), C.grad.numpy(),
                           [float(met["loss/policy_gradient_loss"]) * obs.shape[0] * obs.shape[1] * scale]])


def _lstm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec, p, cs, cp, obs, act, logp, ret, adv, done, c0, h0 = _lstm_problem()
        NGl = obs.shape[1]
        sl = slice(rank * NGl // world, (rank + 1) * NGl // world)       # this rank's env columns = its share of the minibatch
        a = adv[:, sl]
        stats = torch.tensor([a.sum(), (a * a).sum(), a.size], dtype=torch.float64)
        dist.all_reduce(stats)                                           # global advantage statistics
        mean = stats[0].item() / stats[2].item()
        std = np.sqrt(max(stats[1].item() / stats[2].item() - mean * mean, 0.0))
        advn = (a - mean) / (std + 1e-8)
        flat = torch.from_numpy(_lstm_grads(spec, p, cs, cp, obs[:, sl], act[:, sl], logp[:, sl], ret[:, sl], advn, done[:, sl],
                                            c0[sl], h0[sl], 1.0 / obs[:, :, 0].size))
        dist.all_reduce(flat)                                            # ONE collective for the gradients (per network in the library)
        if rank == 0:
            q.put(flat.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_recurrent_update_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lstm_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    flat = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    spec, p, cs, cp, obs, act, logp, ret, adv, done, c0, h0 = _lstm_problem()
    advn = oppo.normalize_advantages(adv.reshape(-1)).reshape(adv.shape)
    exp = _lstm_grads(spec, p, cs, cp, obs, act, logp, ret, advn, done, c0, h0, 1.0 / adv.size)
    np.testing.assert_allclose(flat, exp, rtol=1e-9, atol=1e-12)
