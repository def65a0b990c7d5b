"""Oracle (test infrastructure): CPU restatement of the index sharding of the data-parallel PPO update --
the checker of rl-x_amd/csrc/dist.hip::k_compact_local (rlx_dist_local_rows_i32), never imported by the product.

The minibatch permutation is computed over the GLOBAL flattened index i = t*N_global + n
(rl_x/algorithms/ppo/flax/ppo.py:180-184); rank g owns envs [off, off + N_local).  For every
global minibatch this keeps the rows that live on this rank (order preserved) and rewrites
them to LOCAL flattened indices t*N_local + (n - off).  Pure index plumbing on torch tensors
(works on CPU tensors too -- that is how the gloo tests exercise it)."""
import torch


def local_rows(perm, n_minibatches, minibatch_size, n_global, n_local, env_off):
    """Asynchronous half (pure elementwise work, no host sync): per global index whether it lives on this rank and
    its LOCAL flattened index.  perm: int32[n_minibatches * minibatch_size].  Returns (mask, local) [n_mb, mb]."""
    p = perm.view(n_minibatches, minibatch_size)
    n = p % n_global
    t = p // n_global
    mask = (n >= env_off) & (n < env_off + n_local)
    local = t * n_local + (n - env_off)
    return mask, local


def compact_rows(mask, local):
    """Synchronising half: (compact int32[sum counts] in minibatch order, counts int64[n_mb] CPU, offsets int64[n_mb+1] CPU)."""
    counts = mask.sum(dim=1).cpu()
    compact = local[mask].to(torch.int32).contiguous()
    offsets = torch.zeros(mask.shape[0] + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(counts, 0)
    return compact, counts, offsets


def local_minibatches(perm, n_minibatches, minibatch_size, n_global, n_local, env_off):
    """perm: int32[n_minibatches * minibatch_size] global indices.
    Returns (compact int32[sum counts], counts int64[n_minibatches] (CPU), offsets int64[n+1] (CPU))."""
    mask, local = local_rows(perm, n_minibatches, minibatch_size, n_global, n_local, env_off)
    return compact_rows(mask, local)
