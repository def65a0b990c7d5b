"""GPU: the policy || critic TWIN-launch schedule of rlx_ppo_update_f32 (ppo.hip: twin_fwd_bwd; default for minibatches of at
most 8192 rows -- the per-rank share of BASELINE.json configs[2]) against the two-chain schedule on the same inputs.

Same kernels and tiles per network; the weight-gradient slabs are half as many per network, so gradients agree up to fp32
summation order: losses / gradient norms of the first update to 1e-5, parameters after 8 Adam steps up to the sign flips Adam
makes of gradient entries that are rounding noise.  The gradients of the twin pass are held to the float64 oracle in
tests/test_gpu_bench_shapes.py (twin = 1)."""
import numpy as np
import pytest
import torch

import test_gpu_dist as TD
from rlx_amd.hip import Ctx, PpoHparams
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu


def _run(dev, twin, T, N, E, MB, seed=11, max_norm=5.0, prof=False, tail=-1, l12=1, rec=1):
    ps, cs, pd, cd, P0, C0 = TD._nets(dev, seed=seed)
    S, Ac, LP, R, AD = TD._rollout(dev, T, N, seed=seed)
    hp = PpoHparams(0.1, 0.01, 1.0, max_norm, 0.9, 0.999, 1e-8)
    n_upd = E * (T * N // MB)
    lr = np.linspace(4e-4, 3e-4, n_upd).astype(np.float32)
    c = Ctx(0)
    c.set_option("ppo_twin", twin)
    c.set_option("ppo_tail", tail)
    c.set_option("l12_fused", l12)
    c.set_option("dw_recompute", rec)
    P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
    z = lambda x: torch.zeros_like(x)
    if prof:
        c.prof_begin()
    key, cnt = c.ppo_update(pd, P, z(P), z(P), cd, C, z(C), z(C), S, Ac, LP, R, AD, E, MB, L.prng_key(3), 0, lr, hp, met)
    torch.cuda.synchronize()
    rows = None
    if prof:
        c.prof_end()
        rows = c.prof_rows()
    c.close()
    return P, C, met, key, cnt, rows, (P0, C0)


@pytest.mark.parametrize("T,N,E,MB", [(16, 1024, 2, 4096), (8, 2048, 1, 8192), (16, 1024, 3, 4096)])   # (the last: 12 updates = a group of 8 gathered rows + a short group of 4)
def test_twin_update_matches_the_two_chain_update(dev, T, N, E, MB):
    a = _run(dev, 0, T, N, E, MB)
    b = _run(dev, 1, T, N, E, MB, prof=True)
    n_upd = E * (T * N // MB)
    assert a[4] == b[4] == n_upd and np.array_equal(a[3], b[3])
    ma, mb_ = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert np.all(np.isfinite(mb_))
    # first update: identical parameters and rows -> losses, KL, clip fraction, advantage statistics, gradient norms
    np.testing.assert_allclose(mb_[0, [0, 1, 2, 3, 5, 6, 7]], ma[0, [0, 1, 2, 3, 5, 6, 7]], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(mb_[0, 4], ma[0, 4], rtol=0, atol=1.5 / MB)
    np.testing.assert_allclose(mb_[0, 8:10], ma[0, 8:10], rtol=1e-5)
    # the chain of updates stays together
    np.testing.assert_allclose(mb_[:, [0, 1, 3, 8, 9]], ma[:, [0, 1, 3, 8, 9]], rtol=2e-3, atol=2e-5)
    for x, y, x0 in ((a[0], b[0], a[6][0]), (a[1], b[1], a[6][1])):
        d = (x - y).abs().cpu().numpy()
        ref = x.abs().cpu().numpy()
        assert (d <= 2e-5 + 1e-3 * ref).mean() > 0.995, (d.max(), (d > 2e-5).mean())
        assert d.max() <= 2 * 4e-4 * n_upd
        assert (x - x0).abs().max().item() > 1e-4          # it trained
    # every GEMM row of the twin schedule is ONE launch per update covering both networks
    ran = {(r["kernel"], r["engine"], r["M"], r["N"], r["K"]): r["launches"] for r in b[5]}
    for key in (("k_l12fwd", 1, MB, 256, 512), ("k_tail", 1, MB, 128, 256), ("k_gemm_dw", 1, 768, 384, MB),
                ("k_dx_l1bwd", 1, MB, 512, 256)):
        assert ran.get(key) == n_upd, (key, ran)
    assert not any(r["engine"] == 0 for r in b[5])


def test_twin_update_is_bit_reproducible(dev):
    a = _run(dev, 1, 16, 1024, 2, 4096)
    b = _run(dev, 1, 16, 1024, 2, 4096)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)            # fixed-order reductions, no float atomics: one stream, one result


def test_twin_is_the_default_between_6144_and_16384_rows_only(dev):
    """option ppo_twin = -1 (default): twin launches at 8192-row minibatches, the two-chain schedule at 32768 rows and -- with its
    gathers grouped -- at 4096."""
    tiny = _run(dev, -1, 16, 1024, 1, 4096, prof=True)[5]
    small = _run(dev, -1, 16, 2048, 1, 8192, prof=True)[5]
    large = _run(dev, -1, 16, 4096, 1, 32768, prof=True)[5]
    lt = {(r["kernel"], r["M"], r["N"], r["K"]): r["launches"] for r in tiny}
    ls = {(r["kernel"], r["M"], r["N"], r["K"]): r["launches"] for r in small}
    ll = {(r["kernel"], r["M"], r["N"], r["K"]): r["launches"] for r in large}
    assert lt[("k_l12fwd", 4096, 256, 512)] == 4 * 2      # 4 updates x 2 networks
    assert ls[("k_l12fwd", 8192, 256, 512)] == 4          # 4 updates, one twin launch each
    assert ll[("k_l12fwd", 32768, 256, 512)] == 2 * 2     # 2 updates x 2 networks


def test_default_tail_form_is_64_rows_only_between_8192_and_16384_rows(dev):
    """option ppo_tail = -1 (default) picks k_tail32_bx for small minibatches (launch-latency regime) and above 16384 rows (two-chain
    schedule: fewer HBM bytes beside the other chain's kernels), k_tail_bx in between: the results are bit-identical to the forced form
    (fixed-order arithmetic), and differ from the other form's only at rounding level."""
    for T, N, MB, chosen in ((16, 1024, 4096, 2), (16, 2048, 16384, 1), (16, 4096, 32768, 2)):
        d = _run(dev, -1, T, N, 1, MB, tail=-1)
        f = _run(dev, -1, T, N, 1, MB, tail=chosen)
        o = _run(dev, -1, T, N, 1, MB, tail=3 - chosen)
        assert torch.equal(d[0], f[0]) and torch.equal(d[1], f[1]) and torch.equal(d[2], f[2])
        assert not torch.equal(d[2], o[2])
        np.testing.assert_allclose(d[2][0].cpu().numpy(), o[2][0].cpu().numpy(), rtol=5e-6, atol=1.5 / MB)


def test_a_non_finite_gradient_skips_the_optimizer_step(dev):
    """A NaN in the rollout makes the loss and every gradient non-finite.  The clip + Adam kernels then leave parameters and
    moments untouched (optim.hip: clip_adam_body) and report the non-finite norm: the plugin's per-iteration check raises with
    the last good parameters intact (NaN must not be written over a good state before it is noticed)."""
    T, N, E, MB = 16, 1024, 1, 4096
    for twin in (0, 1):
        ps, cs, pd, cd, P0, C0 = TD._nets(dev, seed=2)
        S, Ac, LP, R, AD = TD._rollout(dev, T, N, seed=2)
        S[3, 5, 2] = float("nan")
        hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
        n_upd = E * (T * N // MB)
        c = Ctx(0)
        c.set_option("ppo_twin", twin)
        P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
        pm, pv, cm, cv = (torch.zeros_like(x) for x in (P, P, C, C))
        c.ppo_update(pd, P, pm, pv, cd, C, cm, cv, S, Ac, LP, R, AD, E, MB, L.prng_key(3), 0, np.full(n_upd, 4e-4, np.float32), hp, met)
        torch.cuda.synchronize()
        m = met.cpu().numpy()
        assert not np.all(np.isfinite(m[:, 8:10]))                     # the poisoned update reports its non-finite norms
        assert torch.isfinite(P).all() and torch.isfinite(C).all()      # ... and never reaches the parameters
        assert torch.isfinite(pm).all() and torch.isfinite(pv).all() and torch.isfinite(cm).all() and torch.isfinite(cv).all()
        assert (P - P0).abs().max().item() > 0                          # the clean updates of the call still stepped
        c.close()


@pytest.mark.parametrize("twin,form", [(0, 1), (1, 1), (0, 2), (1, 2)])
def test_tail_kernel_update_matches_the_three_launch_update(dev, twin, form):
    """k_tail_bx (last hidden layer forward + head + loss + dZ3 + dZ2 in one launch per network) against the launches it replaces
    (k_gemm_bx<0>, k_head_loss_fast, k_gemm_bx<1>), whole updates on both schedules: the same arithmetic per element -- the
    forward product, the head and the input gradient accumulate in the same order -- so the first update's metrics agree to fp32
    rounding and the chain of updates stays together."""
    T, N, E, MB = 16, 1024, 2, 4096
    a = _run(dev, twin, T, N, E, MB, tail=0)
    b = _run(dev, twin, T, N, E, MB, tail=form, prof=True)
    n_upd = E * (T * N // MB)
    assert a[4] == b[4] == n_upd and np.array_equal(a[3], b[3])
    ma, mb_ = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert np.all(np.isfinite(mb_))
    # (form 2 folds a row's head products over eight slices instead of four and rebuilds act'(H2) from fp16 planes: rounding-level differences)
    np.testing.assert_allclose(mb_[0, [0, 1, 2, 3, 5, 6, 7, 8, 9]], ma[0, [0, 1, 2, 3, 5, 6, 7, 8, 9]], rtol=2e-6 if form == 1 else 5e-6, atol=1e-7)
    np.testing.assert_allclose(mb_[0, 4], ma[0, 4], rtol=0, atol=1.5 / MB)
    np.testing.assert_allclose(mb_[:, [0, 1, 3, 8, 9]], ma[:, [0, 1, 3, 8, 9]], rtol=2e-3, atol=2e-5)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        d = (x - y).abs().cpu().numpy()
        ref = x.abs().cpu().numpy()
        assert (d <= 2e-5 + 1e-3 * ref).mean() > 0.995, (d.max(), (d > 2e-5).mean())
    ran = {(r["kernel"], r["engine"], r["M"], r["N"], r["K"]): r["launches"] for r in b[5]}
    assert ran.get(("k_tail", 1, MB, 128, 256)) == n_upd * (1 if twin else 2), ran
    assert ("k_gemm_fwd", 1, MB, 128, 256) not in ran and ("k_gemm_dx", 1, MB, 256, 128) not in ran


@pytest.mark.parametrize("twin", [0, 1])
def test_fused_first_two_layers_match_the_two_launches(dev, twin):
    """k_l12fwd (first + second layer forward in one launch; the first layer on the fp16 pipe like the backward's recompute)
    against k_l1fwd_mfma (exact-fp32 MFMA) + k_gemm_bx<0>: the first-layer products differ by the split engine's 2^-22."""
    T, N, E, MB = 16, 1024, 2, 4096
    a = _run(dev, twin, T, N, E, MB, l12=0)
    b = _run(dev, twin, T, N, E, MB, l12=1)
    ma, mb_ = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert np.all(np.isfinite(mb_))
    np.testing.assert_allclose(mb_[0, [0, 1, 2, 3, 5, 6, 7, 8, 9]], ma[0, [0, 1, 2, 3, 5, 6, 7, 8, 9]], rtol=5e-6, atol=1e-7)
    np.testing.assert_allclose(mb_[:, [0, 1, 3, 8, 9]], ma[:, [0, 1, 3, 8, 9]], rtol=2e-3, atol=2e-5)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        d = (x - y).abs().cpu().numpy()
        ref = x.abs().cpu().numpy()
        assert (d <= 2e-5 + 1e-3 * ref).mean() > 0.995, (d.max(), (d > 2e-5).mean())


@pytest.mark.parametrize("twin,T,N,E,MB", [(0, 16, 1024, 2, 4096), (1, 16, 1024, 2, 4096), (0, 8, 4096, 1, 32768)])
def test_recomputed_first_layer_activations_match_the_stored_ones(dev, twin, T, N, E, MB):
    """dw_recompute = 1 (default): k_l12fwd does not store the [M, 512] first-layer activations; it leaves the rows' LayerNorm
    mean and 1 / std, and the layer-2 weight-gradient job rebuilds its operand (same z1 arithmetic as the forward: observation
    planes x first-layer weight fragments on the fp16 pipe) -- against dw_recompute = 0 (activations stored and read back).
    The rows of a 32-row stage are contracted in another order there (the accumulator layout's), so the layer-2 weight
    gradients agree up to fp32 summation order; everything else is identical."""
    a = _run(dev, twin, T, N, E, MB, rec=0)
    b = _run(dev, twin, T, N, E, MB, rec=1)
    ma, mb_ = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert np.all(np.isfinite(mb_))
    np.testing.assert_allclose(mb_[0, [0, 1, 2, 3, 5, 6, 7]], ma[0, [0, 1, 2, 3, 5, 6, 7]], rtol=1e-6, atol=1e-7)   # forward: same values
    np.testing.assert_allclose(mb_[0, 8:10], ma[0, 8:10], rtol=2e-6)                                            # gradient norms
    np.testing.assert_allclose(mb_[:, [0, 1, 3, 8, 9]], ma[:, [0, 1, 3, 8, 9]], rtol=2e-3, atol=2e-5)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        d = (x - y).abs().cpu().numpy()
        ref = x.abs().cpu().numpy()
        assert (d <= 2e-5 + 1e-3 * ref).mean() > 0.995, (d.max(), (d > 2e-5).mean())
