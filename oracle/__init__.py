"""CPU oracle for the RL-X PPO/SAC hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the algorithm the
reference implements in `rl_x/algorithms/{ppo,sac,ppo_lstm}/flax*/` plus the
third-party semantics it leans on (JAX threefry PRNG, optax Adam/clip, flax
Dense/LayerNorm).  It exists to CHECK the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product package (`rl-x_amd/rlx_amd`) must never import from here.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures
 and JAX/Flax/Optax are not installable in the authoring
container, so this oracle cannot be checked against outputs of
the reference itself.  What pins it instead (tests/test_oracle_*.py):
  * Random123 / JAX Threefry-2x32 known-answer vectors,
  * the long-standing documented values of `jax.random.split(PRNGKey(0))` and
    `jax.random.normal(PRNGKey(0), (1,))` (legacy threefry scheme),
  * closed-form identities (GAE on hand-computable sequences, Adam first
    step, Gaussian log-prob/entropy),
  * float64 `torch.autograd` agreement with the manual backward passes.
"""
