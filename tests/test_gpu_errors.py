"""GPU: error behaviour and degenerate inputs of the C ABI (include/rlx_hip.h): every entry point returns a non-zero code
and leaves a message in rlx_last_error() for arguments it cannot serve -- it never falls back to another path -- empty
inputs are no-ops, and the context stays usable after a refused call."""
import ctypes

import numpy as np
import pytest
import torch

from rlx_amd.hip import PpoHparams, mlp_desc
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu
ACT_TANH, ACT_ELU = 0, 1


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_empty_inputs_are_no_ops(ctx, dev):
    lib = ctx.lib
    d = mlp_desc(17, [64, 64], 6, ACT_TANH, False, True)
    params = torch.zeros(lib.rlx_mlp_param_count(ctypes.byref(d)), device=dev)
    x, out = torch.zeros(4, 17, device=dev), torch.full((4, 6), 7.0, device=dev)
    assert lib.rlx_mlp_fwd_f32(ctx.h, ctypes.byref(d), _p(params), _p(x), _p(out), 0, _stream()) == 0     # n = 0 rows
    a = torch.full((4,), 3.0, device=dev)
    assert lib.rlx_gae_f32(ctx.h, _p(a), _p(a), _p(a), _p(a), _p(a), _p(a), 0, 4, 0.99, 0.9, _stream()) == 0   # T = 0
    assert lib.rlx_gae_f32(ctx.h, _p(a), _p(a), _p(a), _p(a), _p(a), _p(a), 4, 0, 0.99, 0.9, _stream()) == 0   # N = 0
    bits = torch.full((4,), 5, dtype=torch.int32, device=dev)
    key = (ctypes.c_uint32 * 2)(1, 2)
    assert lib.rlx_random_bits_u32(ctx.h, key, _p(bits), 0, 1, _stream()) == 0
    assert lib.rlx_normal_f32(ctx.h, key, _p(out), 0, 1, _stream()) == 0
    torch.cuda.synchronize()
    assert bool((out == 7.0).all()) and bool((a == 3.0).all()) and bool((bits == 5).all())                 # nothing was written


@pytest.mark.parametrize("hidden,msg", [([100, 64], "hidden"), ([64, 64, 64, 64], "n_hidden"), ([64, 30], "multiples of 4")])
def test_unsupported_network_shapes_are_refused(ctx, dev, hidden, msg):
    d = mlp_desc(17, hidden, 6, ACT_TANH, False, False)
    x, out = torch.zeros(8, 17, device=dev), torch.zeros(8, 6, device=dev)
    params = torch.zeros(100000, device=dev)
    with pytest.raises(L.RlxError) as e:
        ctx.mlp_fwd(d, params, x, out)
    assert msg in str(e.value) and "rc=" in str(e.value)


def test_bad_arguments_are_refused_and_the_context_survives(ctx, dev):
    O, A, T, N = 17, 6, 4, 64
    pd = mlp_desc(O, [64, 64], A, ACT_TANH, False, True)
    cd = mlp_desc(O, [64, 64], 1, ACT_TANH, False, False)
    n_p, n_c = (ctx.lib.rlx_mlp_param_count(ctypes.byref(d)) for d in (pd, cd))
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    P, C = 0.05 * torch.randn(n_p, device=dev, generator=g), 0.05 * torch.randn(n_c, device=dev, generator=g)
    z = lambda x: torch.zeros_like(x)
    states, actions = torch.randn(T, N, O, device=dev, generator=g), torch.randn(T, N, A, device=dev, generator=g)
    logp, ret, adv = (torch.randn(T, N, device=dev, generator=g) for _ in range(3))
    hp = PpoHparams(0.2, 0.0, 0.5, 0.5, 0.9, 0.999, 1e-8)
    lr = np.full(8, 3e-4, np.float32)
    met = torch.zeros(8, 10, device=dev)
    with pytest.raises(L.RlxError, match="minibatch"):                       # 256 rows are not a multiple of 100
        ctx.ppo_update(pd, P, z(P), z(P), cd, C, z(C), z(C), states, actions, logp, ret, adv, 1, 100, L.prng_key(1), 0, lr, hp, met)
    with pytest.raises(L.RlxError, match="1-based"):
        ctx.clip_adam_step(P, z(P), z(P), z(P), 0, 3e-4, 0.5)
    with pytest.raises(L.RlxError, match="unknown option"):
        ctx.set_option("no_such_option", 1)
    perm = torch.empty(16, dtype=torch.int32, device=dev)
    with pytest.raises(L.RlxError):
        ctx.permutation(L.prng_key(1), perm, 0, 16)                          # zero epochs
    bad_critic = mlp_desc(O, [64, 64], 2, ACT_TANH, False, False)            # a critic must have one output
    idx = torch.arange(64, dtype=torch.int32, device=dev)
    with pytest.raises(L.RlxError, match="critic"):
        ctx.ppo_minibatch_fwd_bwd(pd, P, z(P), bad_critic, torch.zeros(10000, device=dev), torch.zeros(10000, device=dev),
                                  torch.zeros(8, device=dev), states, actions, logp, ret, adv, idx, hp)
    with pytest.raises(L.RlxError):                                           # host tensors are refused by the binding itself
        ctx.mlp_fwd(pd, P.cpu(), states.view(-1, O), torch.zeros(T * N, A, device=dev))
    # the context is still good: a valid update after the refused calls
    key, cnt = ctx.ppo_update(pd, P, z(P), z(P), cd, C, z(C), z(C), states, actions, logp, ret, adv, 2, 64, L.prng_key(1), 0, lr, hp, met)
    torch.cuda.synchronize()
    assert cnt == 8 and bool(torch.isfinite(met).all()) and bool(torch.isfinite(P).all())


def test_fused_rollout_and_recurrent_envelopes(ctx, dev):
    """Shapes outside a fused kernel's envelope are reported by the *_supported query and refused by the entry point."""
    wide_obs = mlp_desc(40, [512, 256, 128], 6, ACT_ELU, False, True)
    wide_c = mlp_desc(40, [512, 256, 128], 1, ACT_ELU, False, False)
    assert not ctx.rollout_step_supported(wide_obs, wide_c)
    ok_p, ok_c = mlp_desc(17, [512, 256, 128], 6, ACT_ELU, True, True), mlp_desc(17, [512, 256, 128], 1, ACT_ELU, True, False)
    assert ctx.rollout_step_supported(ok_p, ok_c)
    d = L.lstm_policy_desc(17, 6, lstm_hidden=128)                            # this build's recurrent kernels hold a 64-wide cell
    N = 32
    buf = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(L.RlxError, match="lstm_hidden"):
        ctx.ppo_lstm_act(d, buf(400000), ok_c, buf(200000), buf(N, 17), buf(N, 128), buf(N, 128), L.prng_key(0), buf(N, 6), None,
                         buf(N), buf(N))


def test_a_weight_outside_the_engine_window_raises_and_keeps_the_last_good_parameters(monkeypatch):
    """gemm_bx.h: weights enter the fp16 pipe times 64, so |w| >= 1023 becomes inf in the weight image and NaN in every product.
    What must happen: the optimizer steps of the poisoned updates are skipped on the device, the
    plugin's per-iteration check raises and names the engine switch, and the parameters are still the finite ones it started
    the iteration with."""
    import sys
    from rlx_amd.runner.runner import Runner
    import rlx_amd.algorithms.ppo.hip.ppo as ppo_mod
    monkeypatch.setattr(sys, "argv", ["experiment.py", "--algorithm.name=ppo.hip", "--environment.name=synthetic.random_obs",
                                      "--runner.mode=train", "--environment.nr_envs=512", "--algorithm.nr_steps=16",
                                      "--algorithm.minibatch_size=4096", "--algorithm.nr_epochs=1",
                                      "--algorithm.total_timesteps=%d" % (512 * 16 * 2)])
    real_alloc = ppo_mod.PPO._alloc_batch
    holder = {}

    def alloc_and_poison(self):
        batch = real_alloc(self)
        w = self.pparams[17 * 512 + 3 * 512 + 5]          # one weight of the policy's second layer (W1[0][5]) -> 2000
        holder["before"] = (self.pparams.clone(), self.cparams.clone())
        w.fill_(2000.0)
        holder["before"][0][17 * 512 + 3 * 512 + 5] = 2000.0
        holder["model"] = self
        return batch
    monkeypatch.setattr(ppo_mod.PPO, "_alloc_batch", alloc_and_poison)
    # (the Runner logs what train() raises instead of propagating it, like the reference's: let it through for the test)
    from rlx_amd.runner import runner as runner_mod
    monkeypatch.setattr(runner_mod.Runner, "_guarded", lambda self, action, envs, cleanup=(): action())
    with pytest.raises(FloatingPointError) as e:
        Runner().run()
    msg = str(e.value)
    assert "RLX_GEMM_BX=0" in msg and "1023" in msg and "SKIPPED" in msg
    m = holder["model"]
    assert bool(torch.isfinite(m.pparams).all()) and bool(torch.isfinite(m.cparams).all())
    assert bool(torch.isfinite(m.pm).all()) and bool(torch.isfinite(m.pv).all())
    assert torch.equal(m.pparams, holder["before"][0])              # every policy step was skipped: nothing moved
    assert bool(torch.isfinite(m.cm).all()) and bool(torch.isfinite(m.cv).all())
