"""CPU restatement of FastSAC's observation normaliser -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Follows rl_x/algorithms/fastsac/pytorch/observation_normalizer.py:
    __init__   :19-23   running_mean 0, running_var 1, running_std_dev 1 (all [1, O]), count 0 (int64)
    normalize  :27-33   optional update, then (obs - running_mean) / (running_std_dev + epsilon)
    _update    :37-53   batch mean / population variance merged with the running statistics; the squared-difference term
                        uses delta2 = batch_mean - running_mean evaluated AFTER running_mean was overwritten (:44-47).

Pinned by tests/golden/reference_obs_norm.npz: outputs of the reference module itself, executed by file path in the
authoring container (tests/golden/make_reference_golden.py)."""
import numpy as np


class ObservationNormalizer:
    def __init__(self, observation_size, dtype=np.float32, epsilon=1e-8):
        self.dtype = dtype
        self.epsilon = dtype(epsilon)
        self.running_mean = np.zeros((1, observation_size), dtype)
        self.running_var = np.ones((1, observation_size), dtype)
        self.running_std_dev = np.ones((1, observation_size), dtype)
        self.count = np.int64(0)

    def update(self, observations):
        obs = np.asarray(observations, self.dtype)
        batch_mean = obs.mean(axis=0, keepdims=True, dtype=self.dtype)
        batch_var = obs.var(axis=0, keepdims=True, dtype=self.dtype)          # population variance (unbiased=False, :39)
        batch_count = obs.shape[0]
        new_count = self.count + batch_count
        delta = batch_mean - self.running_mean
        self.running_mean = (self.running_mean + delta * self.dtype(batch_count) / self.dtype(new_count)).astype(self.dtype)
        delta2 = batch_mean - self.running_mean                               # against the UPDATED mean, as the reference has it
        m_a = self.running_var * self.dtype(self.count)
        m_b = batch_var * self.dtype(batch_count)
        m2 = m_a + m_b + delta2 ** 2 * self.dtype(self.count) * self.dtype(batch_count) / self.dtype(new_count)
        self.running_var = (m2 / self.dtype(new_count)).astype(self.dtype)
        self.running_std_dev = np.sqrt(self.running_var).astype(self.dtype)
        self.count = np.int64(new_count)

    def normalize(self, observations, update=True):
        if update:
            self.update(observations)
        return ((np.asarray(observations, self.dtype) - self.running_mean) / (self.running_std_dev + self.epsilon)).astype(self.dtype)
