"""Oracle (test infrastructure): JAX's threefry PRNG restated in numpy.

The algorithm lives in a third-party dependency that is NOT under
/root/reference: `jax[cpu]<=0.7.2` (reference `pyproject.toml:12-32`, no lock
file).  Restated from the published algorithm (Random123 Threefry-2x32-20 and
`jax/_src/prng.py` / `jax/_src/random.py`); anchored on the reference's call
sites:

  * `jax.random.PRNGKey` / `split`      rl_x/algorithms/ppo/flax/ppo.py:64-65,114
                                         rl_x/algorithms/ppo/flax_full_jit/ppo.py:75-77,116-117,133,228,291
  * `jax.random.normal`                 rl_x/algorithms/ppo/flax/ppo.py:115
  * `jax.random.permutation(axis=1, independent=True)`
                                         rl_x/algorithms/ppo/flax/ppo.py:191-194
  * `jax.random.randint`                rl_x/algorithms/sac/flax_full_jit/sac.py:281-282

Two split / random_bits schemes exist in JAX (`jax_threefry_partitionable`,
default True since JAX 0.5.0).  Both are implemented; `partitionable=True` is
the default here.

PARITY UNPINNED by the reference (it has no tests); pinned by Random123 KATs
and documented JAX values in tests/test_oracle_prng.py.  Those anchors cover the
threefry block function, `split`, `random_bits` and `normal`.  The shuffle built
on top of them (`permutation_rows`: number of sort rounds, key consumption order,
stable sort by 32-bit keys) and `randint` have NO external anchor in this
container -- no JAX to run, no vectors in the reference -- and follow the JAX
source as published; they stay "parity unpinned" until a JAX-generated vector
is available.
"""
import math
import numpy as np

U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << U32(r)) | (x >> U32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds.  All args uint32 (scalars or arrays, broadcast)."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(k0, dtype=U32)
        k1 = np.asarray(k1, dtype=U32)
        x0 = np.asarray(x0, dtype=U32).copy()
        x1 = np.asarray(x1, dtype=U32).copy()
        ks = (k0, k1, k0 ^ k1 ^ U32(0x1BD11BDA))
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + U32(g + 1)
    return x0, x1


def prng_key(seed):
    """jax.random.PRNGKey(seed) -> uint32[2] = [seed >> 32, seed & 0xffffffff]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=U32)


def _iota_2x32(n):
    idx = np.arange(n, dtype=np.uint64)
    return (idx >> np.uint64(32)).astype(U32), (idx & np.uint64(0xFFFFFFFF)).astype(U32)


def _threefry_legacy_flat(key, count):
    """jax `threefry_2x32(key, iota(count))`: odd counts are zero-padded, the
    flat counter array is split in halves (x0 = first half, x1 = second)."""
    counts = np.arange(count, dtype=U32)
    odd = count % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, U32)])
    h = counts.size // 2
    a, b = threefry2x32(key[0], key[1], counts[:h], counts[h:])
    out = np.concatenate([a, b])
    return out[:-1] if odd else out


def split(key, num=2, partitionable=True):
    """jax.random.split(key, num) -> uint32[num, 2]."""
    key = np.asarray(key, dtype=U32)
    if partitionable:
        hi, lo = _iota_2x32(num)
        b1, b2 = threefry2x32(key[0], key[1], hi, lo)
        return np.stack([b1, b2], axis=1)
    return _threefry_legacy_flat(key, 2 * num).reshape(num, 2)


def random_bits(key, shape, partitionable=True):
    """jax `_random_bits(key, 32, shape)` -> uint32[shape]."""
    key = np.asarray(key, dtype=U32)
    n = int(np.prod(shape)) if len(shape) else 1
    if partitionable:
        hi, lo = _iota_2x32(n)
        b1, b2 = threefry2x32(key[0], key[1], hi, lo)
        return (b1 ^ b2).reshape(shape)
    return _threefry_legacy_flat(key, n).reshape(shape)


def _bits_to_unit_float(bits):
    """uniform [0,1): bitcast((bits >> 9) | 0x3f800000) - 1.0  (float32)."""
    fb = (bits >> U32(9)) | U32(0x3F800000)
    return fb.view(np.float32) - np.float32(1.0)


def uniform(key, shape, minval=0.0, maxval=1.0, partitionable=True):
    f = _bits_to_unit_float(random_bits(key, shape, partitionable))
    lo = np.float32(minval)
    hi = np.float32(maxval)
    return np.maximum(lo, (f * (hi - lo) + lo).astype(np.float32))


def erfinv_f32(x):
    """XLA's f32 ErfInv: M. Giles, "Approximating the erfinv function"
    (single precision polynomial, two branches on w = -log1p(-x*x))."""
    x = np.asarray(x, dtype=np.float32)
    w = (-np.log1p((-x * x).astype(np.float32))).astype(np.float32)
    lt = w < np.float32(5.0)
    wa = (w - np.float32(2.5)).astype(np.float32)
    wb = (np.sqrt(np.maximum(w, np.float32(0))) - np.float32(3.0)).astype(np.float32)
    ca = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
          -0.00125372503, -0.00417768164, 0.246640727, 1.50140941]
    cb = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
          -0.0076224613, 0.00943887047, 1.00167406, 2.83297682]
    pa = np.full_like(x, np.float32(ca[0]))
    pb = np.full_like(x, np.float32(cb[0]))
    for c in ca[1:]:
        pa = (np.float32(c) + pa * wa).astype(np.float32)
    for c in cb[1:]:
        pb = (np.float32(c) + pb * wb).astype(np.float32)
    p = np.where(lt, pa, pb)
    out = (p * x).astype(np.float32)
    return np.where(np.abs(x) == 1, np.float32(np.inf) * x, out).astype(np.float32)


def normal(key, shape, partitionable=True):
    """jax.random.normal(key, shape) float32: sqrt(2) * erf_inv(uniform(nextafter(-1,0), 1))."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, lo, 1.0, partitionable)
    return (np.float32(np.sqrt(2)) * erfinv_f32(u)).astype(np.float32)


def shuffle_num_rounds(size):
    """`_shuffle`: ceil(3 * ln(size) / ln(2^32 - 1)) sort rounds."""
    return int(np.ceil(3 * np.log(max(1, size)) / np.log(np.iinfo(np.uint32).max)))


def permutation_rows(key, x, partitionable=True):
    """jax.random.permutation(key, x, axis=1, independent=True) for 2-D int x:
    `num_rounds` rounds of {key, sub = split(key); stable sort each row by
    random_bits(sub, 32, x.shape)}."""
    key = np.asarray(key, dtype=U32)
    x = np.array(x)
    for _ in range(shuffle_num_rounds(x.size)):
        ks = split(key, 2, partitionable)
        key, sub = ks[0], ks[1]
        sk = random_bits(sub, x.shape, partitionable)
        order = np.argsort(sk, axis=1, kind="stable")
        x = np.take_along_axis(x, order, axis=1)
    return x


def ppo_minibatch_indices(key, batch_size, nr_epochs, nr_minibatches, minibatch_size, partitionable=True):
    """rl_x/algorithms/ppo/flax/ppo.py:191-194.  Returns (new_key, int32[E*M, mb])."""
    ks = split(key, 2, partitionable)
    key, sub = ks[0], ks[1]
    idx = np.tile(np.arange(batch_size, dtype=np.int32), (nr_epochs, 1))
    idx = permutation_rows(sub, idx, partitionable)
    return key, idx.reshape(nr_epochs * nr_minibatches, minibatch_size)


def randint(key, shape, minval, maxval, partitionable=True):
    """jax.random.randint(key, shape, minval, maxval) int32 (span < 2^32)."""
    ks = split(key, 2, partitionable)
    hi_bits = random_bits(ks[0], shape, partitionable)
    lo_bits = random_bits(ks[1], shape, partitionable)
    span = U32(int(maxval) - int(minval))
    with np.errstate(over="ignore"):
        # jax: multiplier = ((2^16 % span)^2) % span, all in uint32 (wraps to 0 when span > 2^16)
        mult = U32(65536) % span
        mult = U32(mult * mult) % span
        off = (U32((hi_bits % span) * mult) + (lo_bits % span)) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)
