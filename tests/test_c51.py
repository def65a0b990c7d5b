"""FastSAC's distributional critic step (rl_x/algorithms/fastsac/pytorch/fastsac.py:144-213).

CPU: oracle/c51.py against tests/golden/reference_c51.npz -- outputs of the reference's own closure, executed with tables of logits
in place of the Q networks (tests/golden/make_reference_golden.py::make_c51): loss, extrema of the next value, gradients w.r.t.
the logits; three cases incl. clipped double Q, n-step discounts, clamping at v_min / v_max and atoms landing exactly on a support
point.  GPU: rlx_c51_critic_loss_f32 through the C ABI against the same fixture and the oracle, 1e-5."""
import os

import numpy as np
import pytest

from oracle import c51

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_c51.npz"))


def _case(tag, c, dtype):
    k = "%s_c%d_" % (tag, c)
    g = lambda n: np.asarray(G[k + n], dtype)
    return dict(q1_logits=g("q1"), q2_logits=g("q2"), q1_next_logits=g("q1_target"), q2_next_logits=g("q2_target"), rewards=g("rewards"),
                dones=g("dones"), truncations=g("truncations"), n_steps=g("n_steps"), next_log_probs=g("next_log_probs"),
                alpha=float(G[k + "alpha"][0]), gamma=float(G[k + "gamma"]), v_min=float(G[k + "v_min"]), v_max=float(G[k + "v_max"]),
                clipped_double_q=bool(G[k + "clipped"])), k


@pytest.mark.parametrize("tag,dtype,rtol,atol", [("f64", np.float64, 1e-11, 1e-13), ("f32", np.float32, 2e-5, 2e-7)])
@pytest.mark.parametrize("c", [0, 1, 2])
def test_oracle_matches_the_reference_closure(tag, dtype, rtol, atol, c):
    kw, k = _case(tag, c, dtype)
    out = c51.critic_loss(**kw)
    assert float(out["q_loss"]) == pytest.approx(float(G[k + "q_loss"]), rel=rtol)
    assert float(out["q_min"]) == pytest.approx(float(G[k + "q_min"]), rel=rtol, abs=atol * 10)
    assert float(out["q_max"]) == pytest.approx(float(G[k + "q_max"]), rel=rtol, abs=atol * 10)
    np.testing.assert_allclose(out["d_q1"], G[k + "d_q1"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(out["d_q2"], G[k + "d_q2"], rtol=rtol, atol=atol)
    g = np.sqrt((np.asarray(G[k + "d_q1"], np.float64) ** 2).sum() + (np.asarray(G[k + "d_q2"], np.float64) ** 2).sum())
    assert g == pytest.approx(float(G[k + "grad_norm"]), rel=1e-5)
    np.testing.assert_allclose(out["target1"].sum(axis=1), 1.0, rtol=0, atol=1e-5 if dtype == np.float32 else 1e-12)   # a distribution


def test_projection_properties():
    """mass is conserved, the expectation of the projection is the clamped Bellman backup of the expectation when nothing clamps,
    and a terminal transition puts all mass next to the reward."""
    rng = np.random.default_rng(0)
    B, NA = 64, 101
    logits = rng.standard_normal((B, NA))
    r = rng.standard_normal(B) * 0.5
    zero, one = np.zeros(B), np.ones(B)
    p, v = c51.project(logits, r, zero, zero, one, zero, 0.0, 0.5, -20.0, 20.0)
    z = np.linspace(-20, 20, NA)
    np.testing.assert_allclose(p.sum(1), 1.0, atol=1e-12)
    np.testing.assert_allclose(v, r + 0.5 * (c51.softmax(logits) * z).sum(1), atol=1e-10)
    p, v = c51.project(logits, r, one, zero, one, zero, 0.0, 0.99, -20.0, 20.0)        # done, not truncated: no bootstrap
    np.testing.assert_allclose(v, r, atol=1e-10)
    assert ((p > 0).sum(1) <= 2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("c", [0, 1, 2])
def test_hip_c51_step_matches_reference_and_oracle(c):
    import torch
    from rlx_amd.hip import Ctx
    dev = torch.device("cuda:0")
    ctx = Ctx(0)
    kw, k = _case("f64", c, np.float64)          # float64 reference outputs on inputs that are then fed as float32
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32))).to(dev)
    kw32 = {n: (np.asarray(v, np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for n, v in kw.items()}
    exp = c51.critic_loss(**kw32)                # the oracle in float64 on the float32-rounded inputs: what the kernel is held to
    B, NA = kw["q1_logits"].shape
    d1, d2, out = torch.empty(B, NA, device=dev), torch.empty(B, NA, device=dev), torch.zeros(4, device=dev)
    ctx.c51_critic_loss(t(kw["q1_logits"]), t(kw["q2_logits"]), t(kw["q1_next_logits"]), t(kw["q2_next_logits"]), t(kw["rewards"]),
                        t(kw["dones"]), t(kw["truncations"]), t(kw["n_steps"]), t(kw["next_log_probs"]),
                        t(np.array([np.log(kw["alpha"])])), kw["gamma"], kw["v_min"], kw["v_max"], kw["clipped_double_q"], d1, d2, out)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    e1, e2 = rel(d1.cpu().numpy(), exp["d_q1"]), rel(d2.cpu().numpy(), exp["d_q2"])
    print(f"c51 case {c}: q_loss {abs(o[0] - exp['q_loss']) / exp['q_loss']:.1e}  d_q1 {e1:.1e}  d_q2 {e2:.1e}  "
          f"q_min {abs(o[1] - exp['q_min']):.1e} q_max {abs(o[2] - exp['q_max']):.1e}")
    assert o[0] == pytest.approx(float(exp["q_loss"]), rel=1e-5) and o[0] == pytest.approx(float(G[k + "q_loss"]), rel=2e-5)
    assert o[1] == pytest.approx(float(exp["q_min"]), rel=1e-5, abs=1e-5) and o[2] == pytest.approx(float(exp["q_max"]), rel=1e-5, abs=1e-5)
    assert e1 < 1e-5 and e2 < 1e-5
    assert rel(d1.cpu().numpy(), np.asarray(G[k + "d_q1"], np.float64)) < 2e-5       # the reference closure's own gradient


@pytest.mark.gpu
def test_hip_c51_step_at_fastsac_size_is_reproducible():
    import torch
    from rlx_amd.hip import Ctx
    dev = torch.device("cuda:0")
    ctx = Ctx(0)
    B, NA = 8192, 101                             # FastSAC's batch_size and nr_atoms (fastsac/pytorch/default_config.py)
    g = torch.Generator(device="cpu").manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    q = [r(B, NA) for _ in range(4)]
    rew, nlp = r(B) * 8, r(B) - 2
    dones = (torch.rand(B, generator=g) < 0.2).float().to(dev)
    tr = dones * (torch.rand(B, generator=g) < 0.5).float().to(dev)
    ns = torch.ones(B, device=dev)
    la = torch.tensor([-0.5], device=dev)
    res = []
    for _ in range(2):
        d1, d2, out = torch.empty(B, NA, device=dev), torch.empty(B, NA, device=dev), torch.zeros(4, device=dev)
        ctx.c51_critic_loss(q[0], q[1], q[2], q[3], rew, dones, tr, ns, nlp, la, 0.99, -20.0, 20.0, False, d1, d2, out)
        torch.cuda.synchronize()
        res.append((d1.clone(), d2.clone(), out.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*res))                    # no atomics anywhere: bit-reproducible
    f64 = lambda x: x.cpu().numpy().astype(np.float64)
    exp = c51.critic_loss(f64(q[0]), f64(q[1]), f64(q[2]), f64(q[3]), f64(rew), f64(dones), f64(tr), f64(ns), f64(nlp), float(np.exp(-0.5)),
                          0.99, -20.0, 20.0, False)
    assert res[0][2][0].item() == pytest.approx(float(exp["q_loss"]), rel=1e-5)
    assert np.linalg.norm(res[0][0].cpu().numpy() - exp["d_q1"]) / np.linalg.norm(exp["d_q1"]) < 1e-5
    np.testing.assert_allclose((f64(res[0][0]) * B).sum(axis=1), 0.0, atol=2e-5)   # softmax - target: rows of the gradient sum to zero


@pytest.mark.gpu
@pytest.mark.parametrize("B,NA,clipped", [(1, 2, True), (5, 3, False), (7, 128, True), (66, 65, False)])
def test_hip_c51_step_edge_sizes(B, NA, clipped):
    """smallest / largest supports (2 and 128 atoms), batches that do not fill the last workgroup, a support wider than one lane
    group (65): against the float64 oracle on the same float32 inputs."""
    import torch
    from rlx_amd.hip import Ctx
    dev = torch.device("cuda:0")
    ctx = Ctx(0)
    rng = np.random.default_rng(B * 131 + NA)
    f32 = lambda a: np.asarray(a, np.float32)
    q = [f32(rng.standard_normal((B, NA))) for _ in range(4)]
    rew, nlp = f32(rng.standard_normal(B) * 3), f32(rng.standard_normal(B) - 1)
    dones = f32(rng.random(B) < 0.4)
    tr = f32((rng.random(B) < 0.5) * dones)
    ns = f32(rng.integers(1, 4, B))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d1, d2, out = torch.empty(B, NA, device=dev), torch.empty(B, NA, device=dev), torch.zeros(4, device=dev)
    ctx.c51_critic_loss(t(q[0]), t(q[1]), t(q[2]), t(q[3]), t(rew), t(dones), t(tr), t(ns), t(nlp), t(f32([np.log(0.3)])), 0.97, -4.0, 6.0,
                        clipped, d1, d2, out)
    torch.cuda.synchronize()
    f64 = lambda a: np.asarray(a, np.float64)
    exp = c51.critic_loss(f64(q[0]), f64(q[1]), f64(q[2]), f64(q[3]), f64(rew), f64(dones), f64(tr), f64(ns), f64(nlp),
                          float(np.exp(np.float32(np.log(0.3)))), 0.97, -4.0, 6.0, clipped)
    o = out.cpu().numpy()
    assert o[0] == pytest.approx(float(exp["q_loss"]), rel=2e-5)
    assert o[1] == pytest.approx(float(exp["q_min"]), rel=2e-5, abs=2e-5) and o[2] == pytest.approx(float(exp["q_max"]), rel=2e-5, abs=2e-5)
    for got, e in ((d1, exp["d_q1"]), (d2, exp["d_q2"])):
        assert np.linalg.norm(got.cpu().numpy() - e) / np.linalg.norm(e) < 2e-5
