"""CPU restatement of FastSAC's observation normaliser -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

What it restates (rl_x/algorithms/fastsac/pytorch/observation_normalizer.py):
    state (:19-23)      mean 0, variance 1, std 1 per observation column, sample count 0
    normalise (:27-33)  optionally fold the batch into the statistics first, then (x - mean) / (std + eps)
    fold (:37-53)       n' = n + m;  mean' = mean + (mu_b - mean) m / n';
                        var' = (var n + var_b m + (mu_b - mean')^2 n m / n') / n'      <- (mu_b - mean') uses the UPDATED mean,
                        std' = sqrt(var')                                                  i.e. not the textbook pooled variance
    with mu_b / var_b the batch mean and POPULATION variance (unbiased=False, :39).

Pinned by tests/golden/reference_obs_norm.npz: outputs of the reference module itself, executed by file path in the authoring
container (tests/golden/make_reference_golden.py)."""
import numpy as np


class ObservationNormalizer:
    def __init__(self, observation_size, dtype=np.float64, epsilon=1e-8):
        self.dtype, self.epsilon = dtype, dtype(epsilon)
        self.running_mean = np.zeros((1, observation_size), dtype)
        self.running_var = np.ones((1, observation_size), dtype)
        self.running_std_dev = np.ones((1, observation_size), dtype)
        self.count = np.int64(0)

    def update(self, observations):
        f = self.dtype
        x = np.asarray(observations, f)
        m, n = x.shape[0], int(self.count)
        total = f(n + m)
        mu_b = x.mean(axis=0, keepdims=True, dtype=f)
        var_b = x.var(axis=0, keepdims=True, dtype=f)                         # population variance
        mean_new = (self.running_mean + (mu_b - self.running_mean) * f(m) / total).astype(f)
        gap = mu_b - mean_new                                                 # against the updated mean, as the reference has it
        second_moment = self.running_var * f(n) + var_b * f(m) + gap ** 2 * f(n) * f(m) / total
        self.running_mean = mean_new
        self.running_var = (second_moment / total).astype(f)
        self.running_std_dev = np.sqrt(self.running_var).astype(f)
        self.count = np.int64(n + m)

    def normalize(self, observations, update=True):
        if update:
            self.update(observations)
        x = np.asarray(observations, self.dtype)
        return ((x - self.running_mean) / (self.running_std_dev + self.epsilon)).astype(self.dtype)
