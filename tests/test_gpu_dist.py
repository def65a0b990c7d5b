"""GPU: the data-parallel PPO update inside the library (rlx_ppo_update_dist_f32, rl-x_amd/csrc/dist.hip + ppo.hip).

  * rlx_dist_local_rows_i32 (stable per-minibatch compaction of the global permutation to this rank's rows) is bit-exact
    against its CPU restatement oracle/sharding.py, ragged counts, empty minibatches and capacity clamp included;
  * one rank, no collectives: the dist entry reproduces rlx_ppo_update_f32;
  * BASELINE.json configs[2] PER-RANK workload at full size: rank 3 of 8 of the 32768-env problem (4096 local envs,
    T = 128, global minibatch 32768 -> 128 minibatches per epoch of ~4096 ragged local rows padded to 4608), with the
    all-reduce emulated through the library's hook: for every update the hook substitutes what an 8-rank all-reduce
    delivers, computed two independent ways -- (a) the single-device gradient of the GLOBAL minibatch, (b) on sampled
    updates the explicit sum of all 8 ranks' local contributions (ragged, zero-weight padding) -- and checks (a) == (b);
    the advantage statistics and the metrics go through the same hook;
  * RCCL itself with the one rank a 1-GPU box offers: communicator creation, rlx_allreduce_grads, the dist update with
    its collectives issued on the communicator stream."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nets
from oracle.sharding import local_minibatches
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu

O, A = 17, 6


def _nets(dev, seed=0):
    rng = np.random.default_rng(seed)
    ps, cs = nets.make_spec("B", O, A, True), nets.make_spec("B", O, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.02 * rng.standard_normal(cs.n_params)).astype(np.float32)
    pd = mlp_desc(O, ps.hidden, A, ps.act, True, True)
    cd = mlp_desc(O, cs.hidden, 1, cs.act, True, False)
    return ps, cs, pd, cd, torch.from_numpy(pp).to(dev), torch.from_numpy(cp).to(dev)


def _rollout(dev, T, N, seed=1):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    return r(T, N, O), r(T, N, A), 0.1 * r(T, N) - 8.5, r(T, N), 2 * r(T, N) + 0.5


class _Buf:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _view(ptr, n, dtype, dev):
    return torch.as_tensor(_Buf(ptr, n, "<f8" if dtype else "<f4"), device=dev)


@pytest.mark.parametrize("world,rank,T,NG,mb,cap_full", [(2, 1, 6, 16, 24, True), (8, 3, 16, 256, 1024, False),
                                                         (4, 0, 8, 64, 128, True), (8, 7, 4, 64, 64, True)])
def test_local_rows_kernel_matches_cpu_restatement(ctx, dev, world, rank, T, NG, mb, cap_full):
    rng = np.random.default_rng(world * 10 + rank)
    B = T * NG
    n_mb = 3 * (B // mb)
    perm = np.concatenate([rng.permutation(B) for _ in range(3)]).astype(np.int32)
    nl = NG // world
    cap = mb if cap_full else ctx.dist_row_capacity(mb, nl, NG)
    assert cap % 1 == 0 and 0 < cap <= mb
    lidx = torch.full((n_mb, cap), -7, dtype=torch.int32, device=dev)
    counts = torch.zeros(n_mb, dtype=torch.int32, device=dev)
    ctx.dist_local_rows(torch.from_numpy(perm).to(dev), n_mb, mb, nl, NG, rank * nl, cap, lidx, counts)
    compact, cnt, offsets = local_minibatches(torch.from_numpy(perm), n_mb, mb, NG, nl, rank * nl)
    assert np.array_equal(counts.cpu().numpy(), cnt.numpy().astype(np.int32))
    got = lidx.cpu().numpy()
    for u in range(n_mb):
        c = int(cnt[u])
        assert np.array_equal(got[u, :c], compact[offsets[u]:offsets[u + 1]].numpy())
        assert (got[u, c:] == -7).all()                      # nothing written beyond the count
    assert ctx.dist_overflow_count() == 0


def test_local_rows_capacity_clamp_is_reported(dev):
    c = Ctx(0)
    try:
        perm = torch.arange(64, dtype=torch.int32, device=dev)           # every row belongs to rank 0 of 1
        lidx = torch.zeros(1, 16, dtype=torch.int32, device=dev)
        counts = torch.zeros(1, dtype=torch.int32, device=dev)
        c.dist_local_rows(perm, 1, 64, 8, 8, 0, 16, lidx, counts)        # capacity 16 < 64 local rows
        # outside an update only the rank-local word moves; the all-rank figure comes from the update's statistics all-reduce
        assert int(counts[0]) == 16 and c.dist_overflow_counts() == (0, 1)
        assert c.dist_overflow_counts() == (0, 0)                         # reading resets the counters
        assert np.array_equal(lidx.cpu().numpy()[0], np.arange(16))
    finally:
        c.close()


def test_one_rank_dist_update_equals_fused_update(ctx, dev):
    T, N, mb, E = 16, 512, 2048, 3
    ps, cs, pd, cd, P0, C0 = _nets(dev)
    states, actions, logp, returns, adv = _rollout(dev, T, N)
    hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
    n_upd = E * (T * N // mb)
    lr = np.linspace(4e-4, 1e-4, n_upd).astype(np.float32)
    z = lambda x: torch.zeros_like(x)
    key = L.prng_key(5)
    P1, C1, met1 = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    k1, c1 = ctx.ppo_update(pd, P1, z(P1), z(P1), cd, C1, z(C1), z(C1), states, actions, logp, returns, adv, E, mb, key, 0, lr,
                            hp, met1)
    P2, C2, met2 = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    ctx.ppo_dist_prefetch(key, E, T, N, N, 0, mb)            # the prefetched plumbing is consumed
    k2, c2 = ctx.ppo_update_dist(pd, P2, z(P2), z(P2), cd, C2, z(C2), z(C2), states, actions, logp, returns, adv, N, 0, E, mb,
                                 key, 0, lr, hp, met2)
    torch.cuda.synchronize()
    assert np.array_equal(k1, k2) and c1 == c2 == n_upd
    assert torch.equal(P1, P2) and torch.equal(C1, C2) and torch.equal(met1, met2)     # same kernels, same order
    # a prefetch for ANOTHER key is discarded safely (waits for the stale side-stream work, regenerates in line)
    P3, C3, met3 = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    ctx.ppo_dist_prefetch(L.prng_key(99), E, T, N, N, 0, mb)
    k3, _ = ctx.ppo_update_dist(pd, P3, z(P3), z(P3), cd, C3, z(C3), z(C3), states, actions, logp, returns, adv, N, 0, E, mb,
                                key, 0, lr, hp, met3)
    torch.cuda.synchronize()
    assert np.array_equal(k1, k3) and torch.equal(P1, P3) and torch.equal(met1, met3)


def test_a_peers_capacity_overflow_reaches_every_rank(dev):
    """Rows dropped by ANY rank travel in slot 3 of the per-minibatch statistics records the update all-reduces: a rank that
    did not overflow itself still reports the overflow in the same iteration (a rank raising alone would leave the others
    blocked in the next collective).  The hook plays the peer: it adds 3 dropped rows to every record."""
    T, NG, mb, E = 8, 64, 32, 2
    ps, cs, pd, cd, P0, C0 = _nets(dev, seed=2)
    S, Ac, LP, R, AD = _rollout(dev, T, NG, seed=2)
    hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
    n_upd = E * (T * NG // mb)
    me = Ctx(0)
    me.set_rank(0, 2)
    seen = {"stats": 0}

    def hook(ptr, n, dtype, on_side):
        if dtype == 1:
            buf = _view(ptr, n, 1, dev).view(n_upd, 4)
            assert float(buf[:, 3].abs().sum()) == 0.0          # this rank dropped nothing
            buf[:, 3] += 3.0
            seen["stats"] += 1
    me.set_allreduce_hook(hook)
    try:
        z = lambda x: torch.zeros_like(x)
        P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
        NL = NG // 2
        mine = tuple(x[:, :NL].contiguous() for x in (S, Ac, LP, R, AD))
        me.ppo_update_dist(pd, P, z(P), z(P), cd, C, z(C), z(C), *mine, NG, 0, E, mb, L.prng_key(1), 0,
                           np.full(n_upd, 1e-4, np.float32), hp, met)
        torch.cuda.synchronize()
    finally:
        me.set_allreduce_hook(None)
    assert seen["stats"] == 1
    assert me.dist_overflow_count() == 3 * n_upd                # rows, summed over ranks: the same number on every rank
    assert me.dist_overflow_counts() == (0, 0)                  # reported once, then reset
    me.close()


@pytest.mark.parametrize("world", [2, 4])
def test_emulated_ranks_small_sum_to_single_device(ctx, dev, world):
    """Every rank of a small job runs the dist update with the hook adding the OTHER ranks' contributions (computed by the
    per-phase entry on a second context); all ranks must end with the single-device parameters (fp32 summation order
    aside) and the global metrics.  With 4 ranks the global minibatch has 8 rows: about one local minibatch in ten is EMPTY
    (a rank without rows still takes part in every collective and applies the summed gradient)."""
    T, NG, mb, E = 8, 64, (32 if world == 2 else 8), 2
    ps, cs, pd, cd, P0, C0 = _nets(dev, seed=world)
    S, Ac, LP, R, AD = _rollout(dev, T, NG, seed=world)
    hp = PpoHparams(0.1, 0.01, 0.7, 5.0, 0.9, 0.999, 1e-8)
    n_upd = E * (T * NG // mb)
    lr = np.full(n_upd, 4e-4, np.float32)
    z = lambda x: torch.zeros_like(x)
    key = L.prng_key(3)
    Pr, Cr, metr = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    ctx.ppo_update(pd, Pr, z(Pr), z(Pr), cd, Cr, z(Cr), z(Cr), S, Ac, LP, R, AD, E, mb, key, 0, lr, hp, metr)
    # global permutation (for the emulated other ranks)
    perm = torch.empty(E * T * NG, dtype=torch.int32, device=dev)
    ctx.permutation(key, perm, E, T * NG)
    nl = NG // world
    aux = (Ctx(0), Ctx(0))      # one per chain: the policy hook (caller's stream) and the critic hook (side stream) overlap
    sawzero = False
    try:
        for rank in range(world):
            shards = [tuple(x[:, r * nl:(r + 1) * nl].contiguous() for x in (S, Ac, LP, R, AD)) for r in range(world)]
            rows = [local_minibatches(perm.cpu(), n_upd, mb, NG, nl, r * nl) for r in range(world)]
            sawzero |= any(int(c.min()) == 0 for _, c, _ in rows)
            me = Ctx(0)
            me.set_rank(rank, world)
            P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
            state = {"p": 0, "c": 0, "calls": 0}
            side = me.side_stream()
            a_all = AD.view(-1)[perm.long()].double().view(n_upd, mb)
            stats_g = torch.stack([a_all.sum(1), (a_all * a_all).sum(1), torch.full((n_upd,), float(mb), device=dev,
                                                                                    dtype=torch.float64),
                                   torch.zeros(n_upd, device=dev, dtype=torch.float64)], dim=1).contiguous()

            def hook(ptr, n, dtype, on_side):
                state["calls"] += 1
                if dtype == 1:                                             # advantage sums [n_upd, 4]
                    buf = _view(ptr, n, 1, dev).view(n_upd, 4)
                    torch.cuda.current_stream().synchronize()
                    loc = buf.clone()
                    assert torch.allclose(loc[:, 2], torch.tensor([float(c) for c in rows[rank][1]], device=dev,
                                                                  dtype=torch.float64))
                    buf.copy_(stats_g)
                    return
                if n == n_upd * 10:                                        # metrics: add the other ranks' partial sums
                    buf = _view(ptr, n, 0, dev).view(n_upd, 10)
                    buf += state["other_met"]
                    if rank != 0:     # rank 0 contributes the replicated values (entropy, adv mean / std, policy std) -- also
                        buf[:, [2, 5, 6, 7]] += metr[:, [2, 5, 6, 7]]      # for minibatches it holds no row of (checked on rank 0)
                    return
                which = "c" if on_side else "p"
                u = state[which]
                state[which] += 1
                buf = _view(ptr, n, 0, dev)
                st = side if on_side else torch.cuda.current_stream()
                with torch.cuda.stream(st):
                    for r in range(world):
                        if r == rank:
                            continue
                        comp, cnt, off = rows[r]
                        idx = comp[off[u]:off[u + 1]].to(dev)
                        if idx.numel() == 0:
                            continue
                        g_p, g_c, m = torch.empty(ps.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
                        aux[on_side].ppo_minibatch_fwd_bwd(pd, P, g_p, cd, C, g_c, m, *shards[r], idx, hp, mb_global=mb,
                                                           stats_io=stats_g[u].clone(), phase=2)
                        buf += g_c if on_side else g_p
                        if not on_side:
                            state["other_met"][u, [0, 3, 4]] += m[[0, 3, 4]]
                        else:
                            state["other_met"][u, 1] += m[1]
            state["other_met"] = torch.zeros(n_upd, 10, device=dev)
            me.set_allreduce_hook(hook)
            k2, c2 = me.ppo_update_dist(pd, P, z(P), z(P), cd, C, z(C), z(C), *shards[rank], NG, rank * nl, E, mb, key, 0, lr,
                                        hp, met)
            torch.cuda.synchronize()
            me.set_allreduce_hook(None)
            me.close()
            assert state["p"] == state["c"] == n_upd and state["calls"] == 2 * n_upd + 2
            for got, exp in ((P, Pr), (C, Cr)):
                d = (got - exp).abs()
                assert ((d <= 2e-5 + 1e-3 * exp.abs()).float().mean().item()) > 0.995, d.max().item()
                assert d.max().item() <= 2 * 4e-4 * n_upd
            np.testing.assert_allclose(met[:4, [0, 1, 3, 4]].cpu().numpy(), metr[:4, [0, 1, 3, 4]].cpu().numpy(), rtol=2e-4, atol=2e-6)
            np.testing.assert_allclose(met[:, [2, 5, 6, 7]].cpu().numpy(), metr[:, [2, 5, 6, 7]].cpu().numpy(), rtol=1e-5, atol=1e-6)
    finally:
        aux[0].close()
        aux[1].close()
    if world == 4:
        assert sawzero          # the empty-local-minibatch path was exercised


@pytest.mark.parametrize("twin", [1, 0])
def test_config2_per_rank_full_size(ctx, dev, twin):
    """BASELINE.json configs[2], the workload of ONE rank (rank 3 of 8) at full size; see the module docstring.
    twin = 1 (option ppo_twin = 1): policy || critic twin launches on one stream, ONE all-reduce per update over
    [policy gradients | pad | critic gradients]; twin = 0 (the default at this per-rank minibatch share of 4608 rows since the
    gathers are grouped): the two-chain schedule, one all-reduce per network and update (policy's on the main stream, critic's on
    the side stream)."""
    T, NG, WORLD, RANK, MB, E = 128, 32768, 8, 3, 32768, 10
    NL = NG // WORLD
    ps, cs, pd, cd, P0, C0 = _nets(dev, seed=2)
    S, Ac, LP, R, AD = _rollout(dev, T, NG, seed=2)                       # the GLOBAL rollout (772 MB of fp32)
    hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
    M = T * NG // MB
    n_upd = E * M
    assert n_upd == 1280
    lr = np.full(n_upd, 4e-4, np.float32)
    key = L.prng_key(8)
    z = lambda x: torch.zeros_like(x)
    perm = torch.empty(E * T * NG, dtype=torch.int32, device=dev)
    k_exp = ctx.permutation(key, perm, E, T * NG)
    cap = ctx.dist_row_capacity(MB, NL, NG)
    assert cap == 4608
    shard = lambda r: tuple(x[:, r * NL:(r + 1) * NL].contiguous() for x in (S, Ac, LP, R, AD))
    mine = shard(RANK)
    # expected advantage statistics of every global minibatch
    a_all = AD.view(-1)[perm.long()].double().view(n_upd, MB)
    stats_g = torch.stack([a_all.sum(1), (a_all * a_all).sum(1), torch.full((n_upd,), float(MB), device=dev, dtype=torch.float64),
                           torch.zeros(n_upd, device=dev, dtype=torch.float64)], dim=1).contiguous()
    del a_all
    aux = (Ctx(0), Ctx(0))      # one per chain (the two hooks overlap on two streams)
    me = Ctx(0)
    me.set_rank(RANK, WORLD)
    me.set_option("ppo_twin", 1 if twin else -1)
    side = me.side_stream()
    np4 = (ps.n_params + 3) // 4 * 4
    P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    full_p, full_c = torch.empty(ps.n_params, device=dev), torch.empty(cs.n_params, device=dev)
    full_m = torch.empty(n_upd, 8, device=dev)
    state = {"p": 0, "c": 0, "checked": 0, "local_counts": None, "worst": 0.0, "worst_local": 0.0}
    SAMPLED = {0, 1, 127, 128, 640, 1279}

    def rank_sum(u, which, local):
        """explicit sum over the 8 ranks' local contributions of update u (per-phase entry, second context); `local` = what
        the library computed for THIS rank (ragged rows padded to the capacity) must equal this rank's term of the sum"""
        tot = torch.zeros(cs.n_params if which else ps.n_params, device=dev, dtype=torch.float64)
        rows = perm[u * MB:(u + 1) * MB]
        n, t_ = rows % NG, rows // NG
        for r in range(WORLD):
            msk = (n >= r * NL) & (n < (r + 1) * NL)
            idx = (t_[msk] * NL + (n[msk] - r * NL)).to(torch.int32).contiguous()
            g_p, g_c, m = torch.empty(ps.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
            aux[which].ppo_minibatch_fwd_bwd(pd, P, g_p, cd, C, g_c, m, *shard(r), idx, hp, mb_global=MB,
                                             stats_io=stats_g[u].clone(), phase=2)
            g = g_c if which else g_p
            if r == RANK:
                state["worst_local"] = max(state["worst_local"], ((local - g).norm() / g.norm()).item())
            tot += g.double()
        return tot

    def hook(ptr, n, dtype, on_side):
        if dtype == 1:
            buf = _view(ptr, n, 1, dev).view(n_upd, 4)
            state["local_counts"] = buf[:, 2].clone()
            buf.copy_(stats_g)
            return
        if n == n_upd * 10:
            buf = _view(ptr, n, 0, dev).view(n_upd, 10)
            buf[:, [0, 1, 3, 4]] = full_m[:, [0, 1, 3, 4]]                  # what summing the 8 ranks' partial sums gives
            buf[:, [2, 5, 6, 7]] = full_m[:, [2, 5, 6, 7]]
            return
        if twin:
            assert n == np4 + cs.n_params and not on_side, (n, on_side)     # ONE collective per update, on the update's stream
            whole = _view(ptr, n, 0, dev)
            handle(0, whole[:ps.n_params], torch.cuda.current_stream())
            handle(1, whole[np4:np4 + cs.n_params], torch.cuda.current_stream())
            return
        assert n == (cs.n_params if on_side else ps.n_params)
        handle(1 if on_side else 0, _view(ptr, n, 0, dev), side if on_side else torch.cuda.current_stream())

    def handle(which, buf, st):
        u = state["c" if which else "p"]
        state["c" if which else "p"] += 1
        with torch.cuda.stream(st):
            # (a) single-device gradient of the GLOBAL minibatch with the current parameters (policy: phase 3, critic: 4)
            idx = perm[u * MB:(u + 1) * MB]
            m = torch.empty(8, device=dev)
            if not which:
                aux[0].ppo_minibatch_fwd_bwd(pd, P, full_p, cd, C, None, m, S, Ac, LP, R, AD, idx, hp, mb_global=MB,
                                             stats_io=stats_g[u].clone(), phase=3)
                full_m[u, [0, 2, 3, 4, 5, 6, 7]] = m[[0, 2, 3, 4, 5, 6, 7]]
            else:
                # the critic half needs its own gather: phase 2 on the critic only is not offered, so run the full pair on aux
                g_p = torch.empty(ps.n_params, device=dev)
                aux[1].ppo_minibatch_fwd_bwd(pd, P, g_p, cd, C, full_c, m, S, Ac, LP, R, AD, idx, hp, mb_global=MB,
                                             stats_io=stats_g[u].clone(), phase=2)
                full_m[u, 1] = m[1]
            full = full_c if which else full_p
            if u in SAMPLED:
                tot = rank_sum(u, which, buf)                              # (b)
                err = ((tot - full.double()).norm() / full.double().norm()).item()
                state["worst"] = max(state["worst"], err)
                state.setdefault("errs", []).append((u, which, round(err, 9)))
                # this rank's own contribution is part of that sum: buf (local) + others == tot
                state["checked"] += 1
            buf.copy_(full)

    me.set_allreduce_hook(hook)
    ar0 = me.get_counter("allreduce_calls")
    try:
        k2, cnt = me.ppo_update_dist(pd, P, z(P), z(P), cd, C, z(C), z(C), *mine, NG, RANK * NL, E, MB, key, 0, lr, hp, met)
        torch.cuda.synchronize()
    finally:
        me.set_allreduce_hook(None)
    assert np.array_equal(k2, k_exp) and cnt == n_upd and state["p"] == state["c"] == n_upd
    # collectives per iteration, as the library counts them (the figure bench.py reports as multi_gpu.collectives_per_step):
    # statistics + metrics once, then ONE gradient all-reduce per update in the twin schedule, one per network otherwise
    assert me.get_counter("allreduce_calls") - ar0 == 2 + (1 if twin else 2) * n_upd
    info = {k: v for k, v in state.items() if k != "local_counts"}
    assert state["checked"] == 2 * len(SAMPLED) and state["worst"] < 1e-5 and state["worst_local"] < 1e-5, info
    assert me.dist_overflow_count() == 0
    # ragged local minibatches: counts follow the global permutation exactly
    n_all = (perm % NG).view(n_upd, MB)
    exp_counts = ((n_all >= RANK * NL) & (n_all < (RANK + 1) * NL)).sum(1).double()
    assert torch.equal(state["local_counts"], exp_counts)
    assert exp_counts.min().item() < 4096 < exp_counts.max().item() <= cap
    # reference trajectory: the fused single-device update of the GLOBAL problem
    Pr, Cr, metr = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    ctx.ppo_update(pd, Pr, z(Pr), z(Pr), cd, Cr, z(Cr), z(Cr), S, Ac, LP, R, AD, E, MB, key, 0, lr, hp, metr)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(met).all())
    # first epoch: same losses (parameters have not had time to drift apart through Adam's rounding amplification)
    np.testing.assert_allclose(met[:M, [0, 1]].cpu().numpy(), metr[:M, [0, 1]].cpu().numpy(), rtol=2e-4, atol=2e-6)
    # (gradient norms are replicated values: rank 0 contributes them to the metric all-reduce, rank 3's copy is masked out)
    for got, exp in ((P, Pr), (C, Cr)):
        d = (got - exp).abs()
        assert (d <= 1e-3 + 1e-2 * exp.abs()).float().mean().item() > 0.99, d.max().item()
    me.close()
    aux[0].close()
    aux[1].close()


def test_rccl_single_rank_communicator(ctx, dev):
    """The 1-GPU box offers one rank: the communicator is still a real RCCL communicator (ncclCommInitRank, 1 of 1), and
    the collectives of the update -- advantage sums (fp64), two gradient all-reduces per update, metrics -- are really
    enqueued on its stream, ordered against both chains with events."""
    lib = L.load_library()
    h = ctypes.c_void_p()
    assert lib.rlx_ctx_create_dist(0, 0, 2, None, ctypes.byref(h)) != 0      # world > 1 without an id is refused
    assert lib.rlx_ctx_create_dist(0, 2, 2, None, ctypes.byref(h)) != 0      # rank out of range
    uid = L.nccl_unique_id()
    assert len(uid) == 128
    # ONE RCCL per process: the entry points are bound to the copy torch mapped (never to a second one loaded next to it)
    assert L.rccl_path() == "(global symbols)" or (L.rccl_path().endswith("librccl.so") and "torch" in L.rccl_path()), L.rccl_path()
    c = Ctx(0, rank=0, world=1, unique_id=uid)
    try:
        assert c.rank_world() == (0, 1)
        g = torch.arange(1000, device=dev, dtype=torch.float32)
        c.allreduce_grads(g)                                                   # sum over one rank
        torch.cuda.synchronize()
        assert torch.equal(g, torch.arange(1000, device=dev, dtype=torch.float32))
        T, N, mb, E = 16, 512, 2048, 2
        ps, cs, pd, cd, P0, C0 = _nets(dev)
        states, actions, logp, returns, adv = _rollout(dev, T, N)
        hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
        n_upd = E * (T * N // mb)
        lr = np.full(n_upd, 4e-4, np.float32)
        z = lambda x: torch.zeros_like(x)
        key = L.prng_key(5)
        P1, C1, met1 = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
        ctx.ppo_update(pd, P1, z(P1), z(P1), cd, C1, z(C1), z(C1), states, actions, logp, returns, adv, E, mb, key, 0, lr, hp, met1)
        P2, C2, met2 = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
        k2, c2 = c.ppo_update_dist(pd, P2, z(P2), z(P2), cd, C2, z(C2), z(C2), states, actions, logp, returns, adv, N, 0, E, mb,
                                   key, 0, lr, hp, met2)
        torch.cuda.synchronize()
        assert c2 == n_upd
        # same gradients; the advantage sums / gradient norms are accumulated in a different order than in the fused path
        for got, exp in ((P2, P1), (C2, C1)):
            d = (got - exp).abs()
            assert (d <= 2e-5 + 1e-3 * exp.abs()).float().mean().item() > 0.999 and d.max().item() <= 2 * 4e-4 * n_upd
        np.testing.assert_allclose(met2[:, :8].cpu().numpy(), met1[:, :8].cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(met2[:, 8:].cpu().numpy(), met1[:, 8:].cpu().numpy(), rtol=1e-4)
    finally:
        c.close()
