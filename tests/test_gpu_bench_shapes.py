"""GPU: the shapes bench.py actually runs, checked against the oracle on the engine the bench runs them on.

  * BASELINE.json configs[1]: one PPO minibatch of 32768 rows gathered out of the 128 x 4096 rollout, nets 512-LN-256-128
    ELU, the DEFAULT engine (hidden-layer GEMMs with split-fp32 operands on the fp16 matrix pipe; the profiler rows prove
    which kernels ran) against oracle/ppo.py in float64 on the host AND against the same loss through torch.autograd in
    float64 on the device (two independent evaluations of the reference's formulas): losses 1e-5, gradients
    ||dg|| / ||g|| < 1e-5 per network and per parameter block.
  * the weight-gradient kernel at the update's layer-2 shape (32768 rows, 512 x 256) on a Cauchy-Schwarz scale, 1e-5.
  * BASELINE.json configs[3] at its FULL shape: a replay ring of 1 M transitions x 4096 envs (obs 376, act 17; 3.08 GB,
    64-bit offsets) filled by 300 vector steps of the plugin's own per-step code (acting / env kernels writing straight into
    the ring slot), so the 244-slot ring wraps; ring contents and one sampled batch of 4096 bit-equal to the reference's numpy
    ring restated in oracle/sac.py (rl_x/algorithms/sac/flax/replay_buffer.py:4-38, pinned bit for bit against that class
    executed: tests/test_oracle_reference_pin.py), the update's losses against oracle/sac.py in float64."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo, prng, sac as osac
from rlx_amd.hip import PpoHparams, mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _desc(spec):
    return mlp_desc(spec.in_dim, spec.hidden, spec.out_dim, spec.act, spec.ln_first, spec.has_logstd)


def _kink_entries(spec, params, x, tau):
    """(layer, row, unit, |z| / rms) of every ReLU pre-activation of the batch within tau of zero, relative to the row's RMS
    pre-activation (float64): a unit within fp32 rounding (a few 1e-7 relative for a 256-term fp32 dot product) of its kink has an
    UNDEFINED fp32 gradient -- any fp32 evaluation may land on either side."""
    h, out = x.astype(np.float64), []
    for li, L in enumerate(spec.layers):
        z = h @ params[L["W"]:L["W"] + L["in"] * L["out"]].reshape(L["in"], L["out"]) + params[L["b"]:L["b"] + L["out"]]
        m = np.abs(z) / np.sqrt((z * z).mean(axis=1, keepdims=True))
        out += [(li, int(i), int(j), float(m[i, j])) for i, j in zip(*np.nonzero(m < tau))]
        h = np.maximum(z, 0.0)
    return out


def _blocks(spec):
    """(name, offset, length) of every parameter block of the flat layout."""
    out = []
    for li, L in enumerate(spec.layers):
        out.append((f"W{li}", L["W"], L["in"] * L["out"]))
        out.append((f"b{li}", L["b"], L["out"]))
        if "g" in L:
            out.append((f"ln_scale{li}", L["g"], L["out"]))
            out.append((f"ln_bias{li}", L["be"], L["out"]))
    out.append(("Wh", spec.head["W"], spec.head["in"] * spec.head["out"]))
    out.append(("bh", spec.head["b"], spec.head["out"]))
    if spec.has_logstd:
        out.append(("logstd", spec.logstd, spec.out_dim))
    return out


@pytest.mark.parametrize("MB,twin,obs_scale,tail,l12", [(32768, 0, 1.0, 1, 1), (32768, 0, 1.0, 0, 0), (32768, 0, 1.0, 1, 0),
                                                        (32768, 1, 1.0, 1, 1), (4096, 1, 1.0, 1, 1), (4096, 1, 1.0, 0, 0),
                                                        (4096, 0, 1.0, 1, 1), (4096, 0, 1.0e4, 1, 1), (4096, 1, 1.0e-6, 1, 1),
                                                        (4096, 0, 1.0e4, 0, 0), (32768, 0, 1.0, 2, 1), (4096, 1, 1.0, 2, 1),
                                                        (4096, 0, 1.0e4, 2, 1)])
def test_bench_minibatch_on_the_split_operand_engine_vs_float64_oracle(ctx, dev, MB, twin, obs_scale, tail, l12):
    """twin = 1: the policy || critic twin-launch pass (ppo.hip: twin_fwd_bwd -- the schedule of the whole-update calls for
    minibatches of at most 8192 rows, i.e. the per-rank share of configs[2]) through the same entry, held to the same bar;
    every profiler row is then ONE launch covering both networks.
    tail = 1 (default): the row-tile-local tail kernel (k_tail_bx: last hidden layer forward + head + loss + both input gradients
    in one launch) instead of the layer-3 forward / head / layer-3 input-gradient launches (tail = 0); tail = 2: its 32-row form with
    the H2 tile resident in LDS as fp16 planes (k_tail32_bx: act'(H2) rebuilt from the planes instead of a second HBM read).
    l12 = 1 (default): first + second layer forward in one launch (k_l12fwd) instead of k_l1fwd_mfma + k_gemm_bx<0> (l12 = 0)."""
    """obs_scale: observations of magnitude 1e4 (un-normalised MuJoCo contact forces: x16 would overflow fp16 -> inf) and 1e-6
    (x16 would lose the low plane): the fused first-layer backward scales its observation planes by the device-side max |x| of
    the pass (common.h: x_scale_from_max), so both stay at the float64 oracle's 1e-5 (VERDICT r04 weak #5 / ADVICE r04)."""
    T, N, O, A = 128, 4096, 17, 6
    B = T * N
    rng = np.random.default_rng(20260927)
    ps, cs = nets.make_spec("B", O, A, True), nets.make_spec("B", O, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.05 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    states = (obs_scale * rng.standard_normal((B, O))).astype(np.float32)
    actions = rng.standard_normal((B, A)).astype(np.float32)
    returns = rng.standard_normal(B).astype(np.float32)
    adv = (rng.standard_normal(B) * 2 + 0.3).astype(np.float32)
    idx = rng.permutation(B)[:MB].astype(np.int32)
    logp = np.zeros(B, np.float32)
    mean, _ = nets.forward(ps, pp, states[idx])
    logp[idx] = (oppo.gaussian_log_prob(actions[idx], mean, pp[ps.logstd:ps.logstd + A][None, :])
                 + 0.05 * rng.standard_normal(MB)).astype(np.float32)
    clip, ent, cc = 0.1, 0.01, 0.7
    f64 = lambda a: a.astype(np.float64)
    # ---- oracle, float64 on the host (manual reverse pass of oracle/nets.py)
    madv = oppo.normalize_advantages(f64(adv[idx]))
    loss_e, met_e, gp_e, gc_e = oppo.ppo_loss_and_grads(ps, f64(pp), cs, f64(cp), f64(states[idx]), f64(actions[idx]),
                                                        f64(logp[idx]), f64(returns[idx]), madv, clip, ent, cc)
    # ---- the same loss through torch.autograd, float64 ON THE DEVICE (independent second evaluation)
    loss_t, gp_t, gc_t = oppo.ppo_loss_torch(ps, f64(pp), cs, f64(cp), f64(states[idx]), f64(actions[idx]), f64(logp[idx]),
                                             f64(returns[idx]), madv, clip, ent, cc, device=dev)
    assert abs(loss_t - loss_e) <= 1e-11 * max(1.0, abs(loss_e))
    assert np.linalg.norm(gp_t - gp_e) / np.linalg.norm(gp_e) < 1e-10
    assert np.linalg.norm(gc_t - gc_e) / np.linalg.norm(gc_e) < 1e-10
    # ---- HIP, default engine, with the profiler recording which kernels ran
    hp = PpoHparams(clip, ent, cc, 0.5, 0.9, 0.999, 1e-8)
    pg, cg, met = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.zeros(8, device=dev)
    dev_in = [_t(x, dev) for x in (states, actions, logp, returns, adv, idx)]
    ctx.set_option("ppo_twin", 1 if twin else 0)
    ctx.set_option("ppo_tail", tail)
    ctx.set_option("l12_fused", l12)
    try:
        ctx.prof_begin()
        ctx.ppo_minibatch_fwd_bwd(_desc(ps), _t(pp, dev), pg, _desc(cs), _t(cp, dev), cg, met, *dev_in, hp)
        ctx.prof_end()
    finally:
        ctx.set_option("ppo_twin", -1)
        ctx.set_option("ppo_tail", -1)
        ctx.set_option("l12_fused", 1)
    rows = ctx.prof_rows()
    ran = {(r["kernel"], r["engine"], r["M"], r["N"], r["K"]): r["launches"] for r in rows}
    # both nets: forward L2 / L3, input gradient L3, weight gradients L3 / L2 on the fp16 pipe; the fused first-layer backward too
    l3 = (("k_tail", 1, MB, 128, 256),) if tail else (("k_gemm_fwd", 1, MB, 128, 256), ("k_gemm_dx", 1, MB, 256, 128))
    l2 = (("k_l12fwd", 1, MB, 256, 512),) if l12 else (("k_gemm_fwd", 1, MB, 256, 512),)
    # (with the tail kernel both upper layers' weight gradients are ONE two-job launch: row keyed by the summed shapes)
    dw = (("k_gemm_dw", 1, 768, 384, MB),) if tail else (("k_gemm_dw", 1, 256, 128, MB), ("k_gemm_dw", 1, 512, 256, MB))
    for key in (("k_dx_l1bwd", 1, MB, 512, 256),) + dw + l2 + l3:
        assert ran.get(key) == (1 if twin else 2), (key, ran)
    assert (("k_gemm_fwd", 1, MB, 128, 256) in ran) == (not tail) and (("k_gemm_fwd", 1, MB, 256, 512) in ran) == (not l12)
    assert not any(r["engine"] == 0 for r in rows), rows
    m = met.cpu().numpy()
    np.testing.assert_allclose(m[0], met_e["loss/policy_gradient_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m[1], met_e["loss/critic_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m[2], met_e["loss/entropy_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m[3], met_e["policy_ratio/approx_kl"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(m[4], met_e["policy_ratio/clip_fraction"], rtol=0, atol=1.5 / MB)
    np.testing.assert_allclose(m[0] - ent * m[2] + cc * m[1], loss_e, rtol=1e-5, atol=1e-6)
    report = {}
    for name, got, exp, spec in (("policy", pg.cpu().numpy(), gp_e, ps), ("critic", cg.cpu().numpy(), gc_e, cs)):
        rel = np.linalg.norm(got - exp) / np.linalg.norm(exp)
        report[name] = rel
        assert rel < 1e-5, (name, rel)
        for bname, off, ln in _blocks(spec):
            e = exp[off:off + ln]
            relb = np.linalg.norm(got[off:off + ln] - e) / max(np.linalg.norm(e), 1e-30)
            report[f"{name}.{bname}"] = relb
            assert relb < 1e-5, (name, bname, relb)
    print("bench-shape minibatch, split-operand engine, ||dg||/||g|| vs float64:", {k: float(f"{v:.2e}") for k, v in report.items()})


def test_weight_gradient_kernel_at_the_layer2_bench_shape(ctx, dev):
    """k_gemm_dw_bx at (32768 rows, Kd 512, N 256): |error| <= 1e-5 |h_col| |dz_col| element-wise (Cauchy-Schwarz scale),
    ||error||_F / ||dW||_F < 1e-5, and within 1.5x of the exact-fp32 engine's own float64-referenced error."""
    M, N, K = 32768, 256, 512
    rng = np.random.default_rng(7)
    Hp = np.where(rng.random((M, K)) < 0.5, rng.standard_normal((M, K)), np.expm1(-np.abs(rng.standard_normal((M, K))))).astype(np.float32)
    dZ = (rng.standard_normal((M, N)) * np.exp(rng.uniform(-6, 0, (M, 1))) / M).astype(np.float32)
    ref = Hp.astype(np.float64).T @ dZ.astype(np.float64)
    scale = np.sqrt((Hp.astype(np.float64) ** 2).sum(0))[:, None] * np.sqrt((dZ.astype(np.float64) ** 2).sum(0))[None, :]
    err = {}
    for mode in (2, 5):
        C, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
        ctx.set_option("bx_gscale_log2", 18)          # = bx_grad_scale(32768): what the minibatch pass of this shape uses
        try:
            ctx.dbg_gemm(mode, _t(Hp, dev), _t(dZ, dev), C, db, M, N, K, 0)
        finally:
            ctx.set_option("bx_gscale_log2", 0)
        e = np.abs(C.cpu().numpy().astype(np.float64) - ref)
        err[mode] = ((e / scale).max(), np.linalg.norm(e) / np.linalg.norm(ref))
        np.testing.assert_allclose(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5 * np.abs(dZ).sum(0).max())
    print("dW layer-2 shape: (max |e| / CS scale, ||e||_F / ||dW||_F) exact fp32:", err[2], " split fp16 planes:", err[5])
    assert err[5][0] < 1e-5 and err[5][1] < 1e-5, err
    assert err[5][1] <= 1.5 * err[2][1] + 1e-9, err


def test_configs3_full_shape_replay_ring_and_update(dev):
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.sac.hip  # noqa: F401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    NE, O, A, BS, CAPACITY, STEPS, WARM = 4096, 376, 17, 4096, 1_000_000, 300, 4

    def make(direct):
        config = ConfigDict()
        config.runner = runner_cfg("train")
        config.algorithm = get_algorithm_config("sac.hip")
        config.environment = get_environment_config("synthetic.random_obs")
        config.environment.nr_envs, config.environment.obs_dim, config.environment.act_dim = NE, O, A
        config.environment.horizon, config.environment.termination_probability = 50, 0.01
        config.algorithm.batch_size, config.algorithm.buffer_size = BS, CAPACITY
        env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
        m = get_algorithm_model_class("sac.hip")(config, env, env, "/tmp/rlx_c3", None)
        m.direct_replay = direct
        # raw lecun heads give std up to e^2 and saturated tanh actions, where log(1 - tanh(u)^2 + 1e-6) cancels catastrophically in
        # ANY fp32 implementation (tests/test_gpu_sac.py, head_scale 1.0: numpy-fp32 vs float64 differ by 5e-4 already); the
        # policy head is scaled by 0.1 -- on both instances -- so that the float64 oracle is a meaningful 1e-5 reference
        lay = osac.make_specs(O, A, 256)[0].head
        m.pparams[lay["W"]:lay["W"] + lay["in"] * lay["out"]] *= 0.1
        m._alloc()
        state, _ = env.reset()
        gen = torch.Generator(device=m.device)
        gen.manual_seed(3)
        return m, env, state.clone(), gen
    m, env, state, gen = make(True)            # the bench's path: kernels write the ring slot
    g, genv, gstate, ggen = make(False)        # generic path: its five transition tensors feed the numpy ring
    assert m.capacity == CAPACITY // NE == 244 and sum(x.numel() * 4 for x in m.ring) > 3.0e9
    rb = osac.ReplayBuffer(CAPACITY, NE, O, A, np.random.default_rng(int(m.seed)))       # sac/flax/sac.py:59
    captured = []
    orig_add = g.replay_add

    def capture(*args):
        captured[:] = [x.detach().to("cpu", copy=True) for x in args]
        orig_add(*args)
    g.replay_add = capture
    for i in range(STEPS):
        state = m.vector_step(env, state, warmup=i < WARM, gen=gen)
        gstate = g.vector_step(genv, gstate, warmup=i < WARM, gen=ggen)
        rb.add(*[x.numpy() for x in captured])
    torch.cuda.synchronize()
    assert STEPS > m.capacity and m.pos == rb.pos == STEPS % 244 and m.size == rb.size == 244
    for dev_arr, host_arr, name in zip(m.ring, (rb.states, rb.next_states, rb.actions, rb.rewards, rb.terminations),
                                       ("states", "next_states", "actions", "rewards", "terminations")):
        assert np.array_equal(dev_arr.cpu().numpy(), host_arr), name           # the wrapped 1 M-transition ring, bit for bit
    assert float(rb.terminations.sum()) > 0
    # ---- one sampled batch (numpy PCG64 draws like the reference) + the update on it
    ps, qs = osac.make_specs(O, A, 256)
    f = lambda x: x.cpu().numpy().astype(np.float64)
    i1, i2 = rb.sample_indices(BS)
    exp_batch = rb.gather(i1, i2)
    before = [x.clone() for x in (m.pparams, m.qparams, m.qtarget, m.log_alpha)]
    key_before = np.array(m.key, copy=True)
    # The plugin's default (and the bench's SAC line): the update gathers from the ring WITHOUT handing the sampled observation
    # rows back (rlx_sac_update_f32 with states = next_states = NULL) -- this is the path whose losses / gradients meet the
    # float64 oracle below.  The generic instance g then replays the SAME draw with the rows requested: the gathered batch is the
    # numpy ring's, bit for bit, and its update equals the elided one in every output.
    assert m.batch_states is False
    m.sample_and_update()
    torch.cuda.synchronize()
    assert np.array_equal(m.idx1.cpu().numpy(), i1.astype(np.int32)) and np.array_equal(m.idx2.cpu().numpy(), i2.astype(np.int32))
    for got, exp in zip(m.batch[2:], exp_batch[2:]):
        assert np.array_equal(got.cpu().numpy(), exp.astype(np.float32))
    assert not m.batch[0].any() and not m.batch[1].any()       # not requested: left as allocated (zeros), never garbage
    g.batch_states = True
    g.rng.bit_generator.state = np.random.default_rng(int(m.seed)).bit_generator.state      # g draws what m / the numpy ring drew
    g._idx_cache = None
    for dst, src in zip((g.pparams, g.qparams, g.qtarget, g.log_alpha), before):
        dst.copy_(src)
    g.parameters_written()
    for name in ("pm", "pv", "qm", "qv", "am", "av"):
        getattr(g, name).zero_()
    g.key, g.opt_count = np.array(key_before, copy=True), 0
    assert m.opt_count == 1
    g.sample_and_update()
    torch.cuda.synchronize()
    assert np.array_equal(g.idx1.cpu().numpy(), i1.astype(np.int32)) and np.array_equal(g.idx2.cpu().numpy(), i2.astype(np.int32))
    for got, exp in zip(g.batch, exp_batch):
        assert np.array_equal(got.cpu().numpy(), exp.astype(np.float32))
    for name in ("pparams", "qparams", "qtarget", "log_alpha", "pm", "pv", "qm", "qv", "am", "av", "metrics_dev"):
        assert torch.equal(getattr(m, name), getattr(g, name)), name
    assert np.array_equal(m.key, g.key)
    new_key_e, e1, e2 = osac.sample_noise(key_before, BS, A, bool(m.scheme))
    s, s2, a, r, term = (x.astype(np.float64) for x in exp_batch)
    met_e, gp_e, gq_e, ga_e = osac.loss_and_grads(ps, f(before[0]), qs, f(before[1]), f(before[2]), np.float64(before[3].item()),
                                                 s, s2, a, r, term, e1.astype(np.float64), e2.astype(np.float64), m.gamma,
                                                 m.target_entropy)
    assert np.array_equal(m.key, new_key_e)
    got = m.metrics_dev.cpu().numpy()
    names = ["loss/q_loss", "loss/policy_loss", "loss/entropy_loss", "entropy/entropy", "entropy/alpha", "q_value/q_value"]
    errs = {n: abs(got[i] - float(met_e[n])) / max(abs(float(met_e[n])), 1e-3) for i, n in enumerate(names)}
    print("configs[3] full-shape update vs float64 oracle (relative):", {k: float(f"{v:.2e}") for k, v in errs.items()})
    for n, e in errs.items():
        assert e < 1e-5, (n, e)
    # ---- gradients.  The 1e-5 bar is defined where the fp32 gradient is: a ReLU unit whose pre-activation sits within fp32
    # rounding of zero for some sample may be "on" in one fp32 evaluation and "off" in another, and that sample's whole backward
    # contribution through the unit flips with it (seen on this very batch: ONE layer-2 unit of critic 0 -> errors of 3e-5 .. 8e-4
    # in exactly b1 / W1 / b0 / W0 of that critic, 2e-7 in every other block).  Such units are found from the FLOAT64
    # pre-activations alone (|z| < 8e-6 of the row's RMS; a few dozen per batch at 393 k unit-samples, closest first); for each, the oracle's
    # gradient with that one unit-sample inverted in the reverse pass (oracle.sac.loss_and_grads: relu_toggle) gives the admissible
    # alternative, and since samples contribute independently the alternatives add.  The device result has to agree to 1e-5 with
    # the oracle for SOME on/off assignment of those few units, and to 1e-5 in every block they do not reach.
    # (Round 4, split-operand engine: critic 1's layer-0 unit 68 of row 2971 on the policy's action sits at |z| / rms = 1.9e-7 and
    # comes out on the other side -- 2.1e-5 in pi.W0, 4.7e-6 over the policy, 2.2e-7 once that one unit-sample is inverted; the
    # exact-fp32 engine, RLX_GEMM_BX=0, happens to agree with float64 on it.  Candidates are therefore tried down to 1e-6.)
    args64 = (ps, f(before[0]), qs, f(before[1]), f(before[2]), np.float64(before[3].item()), s, s2, a, r, term,
              e1.astype(np.float64), e2.astype(np.float64), m.gamma, m.target_entropy)
    n = qs.n_params
    cm, cls, _, _ = osac.policy_forward(ps, f(before[0]), s, m.log_std_min, m.log_std_max)
    ca, _ = osac.tanh_gaussian(cm, cls, e2.astype(np.float64))
    KINK = 8e-6      # |z| / rms below which an fp32 evaluation of a 256-term dot product may land on either side of zero (K eps / 4)
    kinks = []
    for path, spec, par, x in (("q0", qs, f(before[1])[:n], np.concatenate([s, a], 1)), ("q1", qs, f(before[1])[n:], np.concatenate([s, a], 1)),
                               ("qa0", qs, f(before[1])[:n], np.concatenate([s, ca], 1)), ("qa1", qs, f(before[1])[n:], np.concatenate([s, ca], 1)),
                               ("pi", ps, f(before[0]), s)):
        kinks += [(path,) + k for k in _kink_entries(spec, par, x, KINK)]
    # the same for min(Q1, Q2) of the policy loss: a row whose two critics agree to within fp32 rounding follows either critic
    qa = [osac.q_forward(qs, f(before[1]), k, s, ca)[0] for k in range(2)]
    tie = np.abs(qa[0] - qa[1]) / np.sqrt(0.5 * (qa[0] ** 2 + qa[1] ** 2).mean())
    kinks += [("min", 0, int(i), 0, float(tie[i])) for i in np.nonzero(tie < KINK)[0]]
    kinks.sort(key=lambda k: k[4])                         # closest to the kink first
    assert len(kinks) <= 120, len(kinks)
    gp_d, gq_d = m.pm.cpu().numpy() * 10, m.qm.cpu().numpy() * 10
    gp_a, gq_a, taken = gp_e.copy(), gq_e.copy(), []
    rel = lambda d, e: np.linalg.norm(d - e) / max(np.linalg.norm(e), 1e-30)
    for k in kinks:
        critic_path = k[0] in ("q0", "q1")                 # these reach the critics' gradient only; "pi" / "qa*" the policy's only
        if (rel(gq_d, gq_a) if critic_path else rel(gp_d, gp_a)) < 1e-6:
            continue                                       # nothing (left) to explain in that network (fp32 floor: 2e-7 .. 3e-7)
        _, gp_k, gq_k, _ = osac.loss_and_grads(*args64, **({"min_toggle": [k[2]]} if k[0] == "min" else {"relu_toggle": [k[:4]]}))
        if critic_path and rel(gq_d, gq_a + gq_k - gq_e) < 0.9 * rel(gq_d, gq_a):
            gq_a = gq_a + gq_k - gq_e
            taken.append(k)
        elif not critic_path and rel(gp_d, gp_a + gp_k - gp_e) < 0.9 * rel(gp_d, gp_a):
            gp_a = gp_a + gp_k - gp_e
            taken.append(k)
    print(f"configs[3] closest critic tie of the policy loss: |Q1 - Q2| / rms = {tie.min():.1e}")
    print(f"configs[3] units within {KINK:.0e} of the ReLU kink (path, layer, row, unit, |z|/rms): {len(kinks)}; evaluated on the other "
          f"side by the device: {[(k[0], k[1], k[2], k[3], float(f'{k[4]:.1e}')) for k in taken]}")
    assert len(taken) <= 3
    # the fp32 floor of the FORMULA on these inputs (the same restatement evaluated in numpy float32) is printed beside the result
    f32 = lambda x: x.astype(np.float32)
    _, gp_32, gq_32, _ = osac.loss_and_grads(ps, f32(f(before[0])), qs, f32(f(before[1])), f32(f(before[2])), np.float32(before[3].item()),
                                            f32(s), f32(s2), f32(a), f32(r), f32(term), f32(e1), f32(e2), np.float32(m.gamma),
                                            np.float32(m.target_entropy))
    fp, fq = (np.linalg.norm(g32 - g64) / np.linalg.norm(g64) for g32, g64 in ((gp_32, gp_e), (gq_32, gq_e)))
    rp, rq = rel(gp_d, gp_a), rel(gq_d, gq_a)
    blocks = {f"q{k}.{name}": rel(gq_d[k * n + o:k * n + o + ln], gq_a[k * n + o:k * n + o + ln]) for k in range(2) for name, o, ln in _blocks(qs)}
    blocks.update({f"pi.{name}": rel(gp_d[o:o + ln], gp_a[o:o + ln]) for name, o, ln in _blocks(ps)})
    print(f"configs[3] full-shape update: ||dg||/||g|| policy {rp:.2e} (numpy-fp32 formula floor {fp:.2e}; vs the plain oracle "
          f"{rel(gp_d, gp_e):.2e}) critic {rq:.2e} (floor {fq:.2e}; vs the plain oracle {rel(gq_d, gq_e):.2e}); blocks:",
          {k: float(f"{v:.1e}") for k, v in blocks.items()})
    assert rp < 1e-5 and rq < 1e-5 and max(blocks.values()) < 1e-5
