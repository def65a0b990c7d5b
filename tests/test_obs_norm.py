"""FastSAC's observation normaliser (rl_x/algorithms/fastsac/pytorch/observation_normalizer.py).

CPU: oracle/obs_norm.py against tests/golden/reference_obs_norm.npz -- outputs of the reference module itself, executed by
file path by tests/golden/make_reference_golden.py (five batches incl. a one-row batch and constant columns, update / frozen /
eval-mode calls, fp32 and fp64).
GPU: rlx_obs_norm_update_f32 / rlx_obs_norm_apply_f32 through the C ABI against the same fixture and the oracle; the kernel
accumulates in fp64 and rounds once, the bar is 1e-5 relative on every statistic and output (measured ~1e-7)."""
import os

import numpy as np
import pytest

from oracle.obs_norm import ObservationNormalizer

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_obs_norm.npz"))
NB = 5


def _close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,dtype,rtol", [("f64", np.float64, 1e-12), ("f32", np.float32, 2e-5)])
def test_oracle_matches_reference_module(tag, dtype, rtol):
    nrm = ObservationNormalizer(G["batch0"].shape[1], dtype)
    for i in range(NB):
        y = nrm.normalize(G["batch%d" % i], update=True)
        # outputs of constant columns are (x - mean) / (0 + 1e-8): compare them scaled by the denominator
        scale = np.asarray(G["%s_std%d" % (tag, i)], np.float64) + 1e-8
        _close(y * scale, G["%s_out%d" % (tag, i)] * scale, rtol, 1e-4 if tag == "f32" else 1e-9)
        _close(nrm.running_mean, G["%s_mean%d" % (tag, i)], rtol, 1e-6 if tag == "f32" else 1e-12)
        _close(nrm.running_var, G["%s_var%d" % (tag, i)], rtol * 5, 1e-6 if tag == "f32" else 1e-12)
        _close(nrm.running_std_dev, G["%s_std%d" % (tag, i)], rtol * 5, 1e-6 if tag == "f32" else 1e-12)
        assert int(nrm.count) == int(G["%s_count%d" % (tag, i)])
    frozen = nrm.normalize(G["batch2"], update=False)
    _close(frozen, G["%s_frozen" % tag], rtol, 1e-4 if tag == "f32" else 1e-9)
    _close(G["%s_eval" % tag], G["%s_frozen" % tag], 0, 0)                   # eval mode: no update in the reference either
    assert int(nrm.count) == int(G["%s_count_final" % tag]) == sum(G["batch%d" % i].shape[0] for i in range(NB))


def test_reference_quirk_is_kept():
    """delta2 is taken against the already updated mean (observation_normalizer.py:44-47): the merged variance is NOT the exact
    pooled variance.  The fixture (reference output) and the oracle agree with each other and differ from the textbook value."""
    a, b = G["batch0"].astype(np.float64), G["batch2"].astype(np.float64)
    nrm = ObservationNormalizer(a.shape[1], np.float64)
    nrm.update(a)
    nrm.update(b)
    pooled = np.concatenate([a, b]).var(axis=0)
    assert np.max(np.abs(nrm.running_var[0] - pooled) / pooled) > 1e-3


@pytest.mark.gpu
def test_hip_normaliser_matches_reference_and_oracle():
    import torch
    from rlx_amd.hip import Ctx
    dev = torch.device("cuda:0")
    ctx = Ctx(0)
    O = G["batch0"].shape[1]
    mean, var, std = torch.zeros(O, device=dev), torch.ones(O, device=dev), torch.ones(O, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ora = ObservationNormalizer(O, np.float64)
    worst = 0.0
    for i in range(NB):
        x = torch.as_tensor(G["batch%d" % i], device=dev)
        ctx.obs_norm_update(x, mean, var, std, count)
        y = ctx.obs_norm_apply(x, mean, std, torch.empty_like(x))
        ora.normalize(G["batch%d" % i].astype(np.float64), update=True)
        for got, name, o in ((mean, "mean", ora.running_mean), (var, "var", ora.running_var), (std, "std", ora.running_std_dev)):
            g = got.cpu().numpy().astype(np.float64)
            _close(g, G["f64_%s%d" % (name, i)], 1e-5, 1e-6)
            _close(g, o, 1e-5, 1e-6)
            worst = max(worst, float(np.max(np.abs(g - o[0]) / (np.abs(o[0]) + 1e-6))))
        assert int(count.item()) == int(G["f64_count%d" % i])
        # the output against the reference's fp32 run, scaled by the denominator (constant columns divide by 1e-8)
        sc = G["f32_std%d" % i].astype(np.float64) + 1e-8
        _close(y.cpu().numpy() * sc, G["f32_out%d" % i] * sc, 1e-5, 2e-4)
    x = torch.as_tensor(G["batch2"], device=dev)
    y = ctx.obs_norm_apply(x, mean, std, torch.empty_like(x))
    _close(y.cpu().numpy(), G["f64_frozen"], 1e-5, 1e-4)
    ctx.obs_norm_apply(x, mean, std, x)                                         # in place
    assert torch.equal(x, y)
    print("obs norm: worst relative statistic error vs the fp64 oracle %.2e" % worst)


@pytest.mark.gpu
def test_hip_normaliser_large_batch_is_reproducible():
    import torch
    from rlx_amd.hip import Ctx
    dev = torch.device("cuda:0")
    ctx = Ctx(0)
    B, O = 8192, 376                                                            # FastSAC's batch, Humanoid's observation width
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(B, O, generator=g) * 7 + 3).to(dev)
    outs = []
    for _ in range(2):
        mean, var, std = torch.zeros(O, device=dev), torch.ones(O, device=dev), torch.ones(O, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        for k in range(3):
            ctx.obs_norm_update(x[k * 2048:(k + 2) * 2048], mean, var, std, count)
        outs.append((mean.clone(), var.clone(), std.clone(), int(count.item())))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][:3], outs[1][:3])) and outs[0][3] == outs[1][3] == 3 * 4096
    ora = ObservationNormalizer(O, np.float64)
    xn = x.cpu().numpy().astype(np.float64)
    for k in range(3):
        ora.update(xn[k * 2048:(k + 2) * 2048])
    _close(outs[0][0].cpu().numpy(), ora.running_mean, 1e-5, 1e-6)
    _close(outs[0][1].cpu().numpy(), ora.running_var, 1e-5, 1e-6)


@pytest.mark.gpu
def test_sac_plugin_with_observation_normalisation(tmp_path):
    """sac.hip with FastSAC's flag: the statistics follow the sampled batches exactly as fastsac.py:310-311 applies them (states,
    then next states, each with an update), the update sees the normalised rows, acting uses the frozen statistics, and the
    checkpoint carries the normaliser state (fastsac.py:476, :499)."""
    import torch
    from test_gpu_obs_indices import _plugin
    dev = torch.device("cuda:0")
    over = dict(batch_size=64, buffer_size=32 * 64, learning_starts=64, total_timesteps=32 * 12, logging_frequency=32 * 4,
                enable_observation_normalization=True)
    cls, config, env = _plugin("sac.hip", dict(nr_envs=32, obs_dim=40, act_dim=4), over, None, None)
    config.runner.save_model = True
    s = cls(config, env, env, str(tmp_path), None)
    assert s.obs_norm and int(s.norm_count.item()) == 0
    # one hand-driven update: the oracle normaliser over the very rows the ring sampled
    s._alloc()
    state, _ = env.reset()
    for _ in range(4):
        state = s.vector_step(env, state.contiguous(), warmup=True, gen=torch.Generator(device=dev).manual_seed(1))
    seen = {}
    real_update = s.ctx.sac_update

    def spy(*a, **k):
        seen["states"], seen["next"] = a[12][0].clone(), a[12][1].clone()
        return real_update(*a, **k)
    s.ctx.sac_update = spy
    real_sample = s.ctx.sac_replay_sample

    def spy_sample(ring, i1, i2, batch):
        real_sample(ring, i1, i2, batch)
        seen["raw"], seen["raw_next"] = batch[0].clone(), batch[1].clone()
    s.ctx.sac_replay_sample = spy_sample
    s.sample_and_update()
    s.ctx.sac_update, s.ctx.sac_replay_sample = real_update, real_sample
    ora = ObservationNormalizer(40, np.float64)
    e0 = ora.normalize(seen["raw"].cpu().numpy().astype(np.float64), update=True)
    e1 = ora.normalize(seen["raw_next"].cpu().numpy().astype(np.float64), update=True)
    _close(seen["states"].cpu().numpy(), e0, 1e-5, 1e-5)
    _close(seen["next"].cpu().numpy(), e1, 1e-5, 1e-5)
    _close(s.norm_mean.cpu().numpy(), ora.running_mean, 1e-5, 1e-6)
    _close(s.norm_std.cpu().numpy(), ora.running_std_dev, 1e-5, 1e-6)
    assert int(s.norm_count.item()) == 128
    # acting: frozen statistics
    x = torch.randn(32, 40, device=dev) * 3 + 1
    got = s.policy_obs(x).clone()
    _close(got.cpu().numpy(), ora.normalize(x.cpu().numpy().astype(np.float64), update=False), 1e-5, 1e-5)
    assert int(s.norm_count.item()) == 128
    # a short training run and the checkpoint round trip
    s.train()
    assert all(np.isfinite(v) for v in s.last_metrics.values()) and int(s.norm_count.item()) > 128
    s.save()
    config.runner.load_model = s.save_path + "/" + s.best_model_file_name
    r = cls.load(config, env, env, str(tmp_path), None, [])
    assert r.obs_norm and torch.equal(r.norm_mean, s.norm_mean) and torch.equal(r.norm_std, s.norm_std)
    assert int(r.norm_count.item()) == int(s.norm_count.item())


@pytest.mark.gpu
def test_sharded_batches_give_the_one_device_statistics(dev):
    """Data parallel: every rank feeds ITS share of the batch; the library all-reduces the fp64 column sums and the row count
    (one collective of 2 O + 1 doubles), so all ranks merge the same global batch -- the running statistics equal those of one
    device seeing the whole batch, and stay replicated.  Ranks emulated through the all-reduce hook."""
    import torch
    from rlx_amd.hip import Ctx

    class _Buf:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    rng = np.random.default_rng(3)
    O, B, world = 37, 3000, 3
    x = torch.from_numpy((rng.standard_normal((2, B, O)) * rng.uniform(0.1, 5, O) + rng.uniform(-3, 3, O)).astype(np.float32)).to(dev)

    def state():
        return (torch.zeros(O, device=dev), torch.ones(O, device=dev), torch.ones(O, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    whole, one = Ctx(0), state()
    for k in range(2):
        whole.obs_norm_update(x[k], *one)
    shard = B // world
    ranks = [state() for _ in range(world)]
    for k in range(2):
        peers = []
        for rk in range(world - 1, -1, -1):              # the peers first (their contribution is captured), rank 0 last adds them up
            c = Ctx(0)
            c.set_rank(rk, world)

            def hook(ptr, n, dtype, on_side, rk=rk):
                assert dtype == 1 and n == 2 * O + 1
                buf = torch.as_tensor(_Buf(ptr, n), device=dev)
                if rk:
                    peers.append(buf.clone())
                else:
                    for p in peers:
                        buf += p
            c.set_allreduce_hook(hook)
            try:
                c.obs_norm_update(x[k, rk * shard:(rk + 1) * shard].contiguous(), *ranks[rk])
                torch.cuda.synchronize()
            finally:
                c.set_allreduce_hook(None)
                c.close()
        # what rank 0 computed is what every rank computes from the reduced sums: copy it to the emulated peers' state
        for rk in range(1, world):
            for dst, src in zip(ranks[rk], ranks[0]):
                dst.copy_(src)
    for got, exp in zip(ranks[0][:3], one[:3]):
        np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=2e-7, atol=1e-7)
    assert int(ranks[0][3][0]) == int(one[3][0]) == 2 * B
