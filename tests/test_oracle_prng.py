"""CPU: pins the oracle's PRNG restatement with external known-answer vectors
(the reference has no tests of its own)."""
import numpy as np

from oracle import prng


def test_threefry2x32_random123_kats():
    # Random123 kat_vectors, threefry2x32 20 rounds
    cases = [((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
             ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
             ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]
    for key, ctr, exp in cases:
        a, b = prng.threefry2x32(key[0], key[1], ctr[0], ctr[1])
        assert (int(a), int(b)) == exp


def test_split_documented_values_legacy():
    # long-standing JAX docs values (jax_threefry_partitionable=False)
    k = prng.split(prng.prng_key(0), 2, partitionable=False)
    assert k.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    k = prng.split(prng.prng_key(42), 2, partitionable=False)
    assert k.tolist() == [[2465931498, 3679230171], [255383827, 267815257]]


def test_split_documented_values_partitionable():
    # JAX >= 0.5 docs ("Pseudorandom numbers" tutorial): split(key(42))
    k = prng.split(prng.prng_key(42), 2, partitionable=True)
    assert k.tolist() == [[1832780943, 270669613], [64467757, 2916123636]]


def test_normal_documented_values():
    assert np.float32(prng.normal(prng.prng_key(0), (1,), False)[0]) == np.float32(-0.20584226)
    assert np.float32(prng.normal(prng.prng_key(42), (), False)) == np.float32(-0.18471177)
    # JAX >= 0.5 docs: random.normal(key(42)) -> -0.028304616
    assert np.float32(prng.normal(prng.prng_key(42), (), True)) == np.float32(-0.028304616)


def test_legacy_bits_odd_count_padding():
    b3 = prng.random_bits(prng.prng_key(0), (3,), False)
    b4 = prng.random_bits(prng.prng_key(0), (4,), False)
    assert b3.shape == (3,) and b4.shape == (4,)
    # first half of the counter array is [0,1] in both cases
    assert b3[0] == 4146024105  # == split(PRNGKey(0))[0][0]: same (key, counter) pair


def test_shuffle_rounds():
    assert prng.shuffle_num_rounds(10 * 524288) == 3
    assert prng.shuffle_num_rounds(10 * 4194304) == 3
    assert prng.shuffle_num_rounds(10 * 2048) == 2
    assert prng.shuffle_num_rounds(10 * 64) == 1
    assert prng.shuffle_num_rounds(1) == 0


def test_permutation_is_bijection_per_row():
    for scheme in (True, False):
        key, idx = prng.ppo_minibatch_indices(prng.prng_key(1), 96, 3, 4, 24, scheme)
        rows = idx.reshape(3, 96)
        for r in rows:
            assert sorted(r.tolist()) == list(range(96))
        assert not np.array_equal(rows[0], rows[1])
        assert key.dtype == np.uint32


def test_uniform_range_and_randint():
    u = prng.uniform(prng.prng_key(3), (1000,))
    assert u.min() >= 0 and u.max() < 1
    r = prng.randint(prng.prng_key(3), (1000,), 0, 7)
    assert r.min() >= 0 and r.max() < 7 and len(set(r.tolist())) == 7
