"""CPU oracle for the RL-X PPO/SAC hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the algorithm the
reference implements in `rl_x/algorithms/{ppo,sac,ppo_lstm}/flax*/` plus the
third-party semantics it leans on (JAX threefry PRNG, optax Adam/clip, flax
Dense/LayerNorm).  It exists to CHECK the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product package (`rl-x_amd/rlx_amd`) must never import from here.

PINNING STATUS.  The reference has no tests, golden vectors or fixtures
 and JAX/Flax/Optax are not installable in the authoring
container, so the JAX flavours (the parity target) cannot be
run.  Two parts of the reference DO run here and pin the oracle through
committed fixtures (tests/golden/make_reference_golden.py ->
tests/golden/reference_*.npz, checked by tests/test_oracle_reference_pin.py):
  * its PyTorch flavour of the same algorithms (network modules loaded by
    file path, the GAE / loss / optimiser closures of `PPO.train`/`SAC.train`
    compiled from the reference file): Dense+tanh / Dense+relu networks,
    Gaussian and tanh-Gaussian log-probs, GAE, the PPO clipped-surrogate and
    value losses with all gradients, clip + Adam over two steps, the SAC
    critic / policy / alpha losses with all gradients and Adam, the
    Categorical PPO head (oracle/discrete.py)  -- PINNED;
  * the JAX flavour's numpy replay ring (sac/flax/replay_buffer.py) -- PINNED,
    bit for bit.
PARITY UNPINNED for what only exists in JAX/Flax: the threefry PRNG and the
key schedules, `jax.random.permutation`, flax LayerNorm / ELU torso,
OptimizedLSTMCell / GRUCell, the full-jit loop structure.  Those are pinned by
(tests/test_oracle_*.py):
  * Random123 / JAX Threefry-2x32 known-answer vectors,
  * the long-standing documented values of `jax.random.split(PRNGKey(0))` and
    `jax.random.normal(PRNGKey(0), (1,))` (legacy threefry scheme),
  * closed-form identities (GAE on hand-computable sequences, Adam first
    step, Gaussian log-prob/entropy),
  * float64 `torch.autograd` agreement with the manual backward passes.
"""
