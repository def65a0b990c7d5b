"""GPU: threefry bits / normal / permutation vs the oracle.  Integer work is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import prng
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme", [1, 0])
@pytest.mark.parametrize("n", [1, 2, 7, 64, 1001, 70000])
def test_random_bits_bit_exact(ctx, dev, scheme, n):
    key = prng.prng_key(1234)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.random_bits(key, out, scheme)
    got = out.cpu().numpy().view(np.uint32)
    exp = prng.random_bits(key, (n,), partitionable=bool(scheme))
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("scheme", [1, 0])
def test_split_host_matches_oracle(scheme):
    for seed in (0, 1, 42, 2**33 + 5):
        for num in (2, 3, 4, 9):
            got = L.threefry_split(prng.prng_key(seed), num, scheme)
            exp = prng.split(prng.prng_key(seed), num, bool(scheme))
            assert np.array_equal(got, exp)


@pytest.mark.parametrize("scheme", [1, 0])
def test_normal_matches_oracle(ctx, dev, scheme):
    key = prng.prng_key(7)
    n = 4096 * 6
    out = torch.empty(n, dtype=torch.float32, device=dev)
    ctx.normal(key, out, scheme)
    got = out.cpu().numpy()
    exp = prng.normal(key, (n,), bool(scheme))
    # transcendental ulp differences only (log1p/sqrt on device vs numpy)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-6)  # tails: d erfinv/du is large near |u| -> 1
    assert abs(got.mean()) < 0.03 and abs(got.std() - 1) < 0.03


@pytest.mark.parametrize("scheme", [1, 0])
@pytest.mark.parametrize("E,B", [(1, 1), (2, 5), (3, 64), (10, 2048), (4, 20000), (2, 131072),
                                 (1, 600001)])      # (147 tiles: more than the 128 one pass of the scan kernel's lanes covers)
def test_permutation_bit_exact(ctx, dev, scheme, E, B):
    key = prng.prng_key(1)
    out = torch.empty(E * B, dtype=torch.int32, device=dev)
    new_key = ctx.permutation(key, out, E, B, scheme)
    got = out.cpu().numpy().reshape(E, B)
    ks = prng.split(key, 2, bool(scheme))
    exp = prng.permutation_rows(ks[1], np.tile(np.arange(B, dtype=np.int32), (E, 1)), bool(scheme))
    assert np.array_equal(new_key, ks[0])
    assert np.array_equal(got, exp)


def test_permutation_full_size_properties(ctx, dev):
    """BASELINE config 2 size (E=10, B=524288): every row a bijection, rows differ, deterministic."""
    E, B = 10, 524288
    out = torch.empty(E * B, dtype=torch.int32, device=dev)
    ctx.permutation(prng.prng_key(1), out, E, B, 1)
    rows = out.view(E, B)
    srt, _ = torch.sort(rows, dim=1)
    assert torch.equal(srt, torch.arange(B, device=dev, dtype=torch.int32).expand(E, B))
    assert not torch.equal(rows[0], rows[1])
    out2 = torch.empty_like(out)
    ctx.permutation(prng.prng_key(1), out2, E, B, 1)
    assert torch.equal(out, out2)
    # first 4096 entries of row 0 against the oracle's full-size run would take ~10 s on CPU; check
    # instead a checksum-of-checksums that the oracle reproduces for one row.
    k = prng.split(prng.prng_key(1), 2, True)[1]
    exp_row0 = prng.permutation_rows(k, np.tile(np.arange(B, dtype=np.int32), (E, 1)), True)[0]
    assert np.array_equal(rows[0].cpu().numpy(), exp_row0)
