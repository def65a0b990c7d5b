"""CPU: the host side of sac.hip's vector step (rl-x_amd/rlx_amd/algorithms/sac/hip/sac.py::vector_step) with the library
and the env replaced by recording stand-ins -- what goes where in the replay ring, in which order, and what the env receives
(replay_buffer.py:11-28 ring semantics, sac.py:251-262 warm-up, policy.py:44-48 processed action).  The kernels themselves are
checked on the GPU (tests/test_gpu_sac.py); this pins the bookkeeping around them."""
import numpy as np
import pytest
import torch

from oracle import sac as osac
from rlx_amd.algorithms.sac.hip.sac import SAC


class _FakeCtx:
    """sac_act stand-in: action = tanh(obs[:, :A] + call index), processed through the same formula the kernel implements"""

    def __init__(self):
        self.calls = 0

    def sac_act(self, pdesc, pparams, obs, key, action, ls_min, ls_max, scheme=1, processed=None, **kw):
        self.calls += 1
        a = torch.tanh(obs[:, :action.shape[1]] + 0.1 * self.calls)
        action.copy_(a)
        if processed is not None:
            low, half, out = processed
            out.copy_(low + (a.clamp(-1, 1) + 1.0) * half)
        return np.array([key[0] + 1, key[1]], dtype=np.uint32)


class _FakeEnv:
    """device-env stand-in with the `step_into` contract of environments/synthetic/random_obs"""

    def __init__(self, N, O):
        self.N, self.O, self.t = N, O, 0
        self.obs = torch.arange(N * O, dtype=torch.float32).reshape(N, O) * 1e-2
        self.received = []

    def _advance(self, action):
        self.received.append(action.clone())
        self.t += 1
        final = self.obs + 100.0 * self.t                 # what the episode saw last
        reward = action.sum(dim=1)
        term = (torch.arange(self.N) % 3 == self.t % 3).float()
        self.obs = self.obs + 1.0                          # auto-reset envs continue from here
        return final, reward, term

    def step_into(self, action, ring_ns, ring_r, ring_t):
        final, reward, term = self._advance(action)
        ring_ns.copy_(final); ring_r.copy_(reward); ring_t.copy_(term)

    def step(self, action):
        final, reward, term = self._advance(action)
        return self.obs, reward, term > 0.5, torch.zeros(self.N, dtype=torch.bool), {"final_observation": final}


def _model(N, O, A, cap, direct):
    m = object.__new__(SAC)
    m.torch = torch
    m.device = torch.device("cpu")
    m.nr_envs, m.obs_dim, m.act_dim, m.batch_size, m.buffer_size = N, O, A, 4, cap * N
    m.env_as_low = torch.tensor([-2.0, -1.0, 0.5][:A])
    m.env_as_high = torch.tensor([2.0, 3.0, 1.5][:A])
    m.log_std_min, m.log_std_max, m.scheme = -20.0, 2.0, 1
    m.pdesc = m.pparams = None
    m.key = np.array([7, 9], dtype=np.uint32)
    m.ctx = _FakeCtx()
    m.direct_replay = direct
    m._alloc()
    return m


@pytest.mark.parametrize("direct", [True, False])
def test_vector_step_fills_the_replay_ring_like_the_reference_buffer(direct):
    N, O, A, cap = 6, 5, 3, 4
    m = _model(N, O, A, cap, direct)
    env = _FakeEnv(N, O)
    gen = torch.Generator().manual_seed(3)
    state = env.obs.clone()
    ref = osac.ReplayBuffer(cap * N, N, O, A, np.random.default_rng(0))   # replay_buffer.py restated (numpy)
    for i in range(7):                                                   # wraps the 4-slot ring; two warm-up steps
        obs_before = env.obs.clone()
        state = m.vector_step(env, state, warmup=i < 2, gen=gen)
        slot = i % cap
        stored_action = m.ring[2][slot]
        # the env received the processed form of exactly the action that was stored (policy.py:44-48)
        np.testing.assert_allclose(env.received[-1].numpy(),
                                   osac.processed_action(stored_action.numpy().astype(np.float64), m.env_as_low.numpy(),
                                                         m.env_as_high.numpy()), rtol=1e-6, atol=1e-6)
        assert torch.equal(m.ring[0][slot], obs_before)                  # pre-step observation
        assert torch.equal(m.ring[1][slot], obs_before + 100.0 * env.t)  # FINAL observation, not the reset one
        assert torch.equal(state, env.obs)
        ref.add(obs_before.numpy(), m.ring[1][slot].numpy(), stored_action.numpy(), m.ring[3][slot].numpy(), m.ring[4][slot].numpy())
        assert m.pos == ref.pos and m.size == ref.size
        if i < 2:
            assert m.ctx.calls == 0 and float(stored_action.abs().max()) <= 1.0   # uniform warm-up actions in [-1, 1)
    assert m.ctx.calls == 5 and m.key[0] == 7 + 5                         # one key split per policy action
    for got, exp in zip(m.ring, (ref.states, ref.next_states, ref.actions, ref.rewards, ref.terminations)):
        np.testing.assert_array_equal(got.numpy(), exp.astype(np.float32))


def test_direct_and_generic_paths_store_the_same_transitions():
    rings = []
    for direct in (True, False):
        m = _model(6, 5, 3, 4, direct)
        env = _FakeEnv(6, 5)
        gen = torch.Generator().manual_seed(11)
        state = env.obs.clone()
        for i in range(9):
            state = m.vector_step(env, state, warmup=i < 3, gen=gen)
        rings.append([x.clone() for x in m.ring] + [state.clone()])
    for a, b in zip(*rings):
        assert torch.equal(a, b)
