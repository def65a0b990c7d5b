"""CPU: the arithmetic identity behind the split-operand GEMM engine (rl-x_amd/csrc/gemm_bx.h), restated in numpy.

  * an fp32 value (times a power-of-two scale that puts it inside fp16's normal range) splits into two fp16 planes
    a = a0 + a1 + e with |e| <= 2^-22 |a| (round-to-nearest-even conversions; the residual a - a0 is exact in fp32); below the
    normal range the planes keep an absolute resolution of 2^-25;
  * the three plane products a0 b0 + a0 b1 + a1 b0, accumulated in fp32 per 16-k block like the MFMA chain, reproduce a dot
    product with an fp64-referenced error NOT LARGER than that of a sequential fp32 fma chain (the exact-fp32 engine) -- for
    O(1) activations against 0.05-scale weights (forward) and for 1 / minibatch-scale gradients with the pass's gradient scale
    (backward); the single product a0 b0 does not, which is why the kernels issue three MFMAs per 16 k.
The GPU suite checks the kernels themselves (tests/test_gpu_gemm.py, tests/test_gpu_bench_shapes.py); this file pins the
numerical argument on the CPU."""
import numpy as np

W_SCALE = 64.0            # X_WSCALE of gemm_bx.h


def grad_scale(rows):     # bx_grad_scale of gemm_bx.h: 8 * 2^ceil(log2 rows)
    s, r = 8.0, 1
    while r < rows:
        s *= 2.0
        r <<= 1
    return s


def _split2(a, scale=1.0):
    a = (np.asarray(a, np.float32) * np.float32(scale)).astype(np.float32)      # power of two: exact
    a0 = a.astype(np.float16).astype(np.float32)                                 # v_cvt_pk_f16_f32: RNE
    a1 = (a - a0).astype(np.float32).astype(np.float16).astype(np.float32)       # residual exact in fp32
    return a0, a1


def test_two_planes_carry_22_bits_inside_the_normal_range_and_2_pow_minus_25_below():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)   # 3e-4 .. 3e3
    a = a[np.abs(a) < 60000.0]
    a0, a1 = _split2(a)
    err = np.abs(a.astype(np.float64) - (a0.astype(np.float64) + a1.astype(np.float64)))
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(a.astype(np.float64)), 2.0 ** -25))
    tiny = (rng.standard_normal(50000) * 1e-6).astype(np.float32)               # far below fp16's normal range
    t0, t1 = _split2(tiny)
    assert np.all(np.abs(tiny.astype(np.float64) - (t0.astype(np.float64) + t1.astype(np.float64))) <= 2.0 ** -25)
    assert np.isinf(_split2(np.float32(70000.0))[0])                            # beyond the range: inf, loudly


def _dot_products(A, B, pairs, sa, sb):
    Ap, Bp = _split2(A, sa), _split2(B, sb)
    M, K = A.shape
    out = np.zeros((M, B.shape[1]), np.float32)
    for k0 in range(0, K, 16):
        for p, q in pairs:
            out = (out + (Ap[p][:, k0:k0 + 16].astype(np.float64) @ Bp[q][k0:k0 + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return out / np.float32(sa * sb)


def _seq_fp32(A, B):
    seq = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k in range(A.shape[1]):                     # sequential fp32 fma chain = the exact-fp32 MFMA engine
        seq = (seq.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1].astype(np.float64)).astype(np.float32)
    return seq


def test_three_products_match_fp32_accuracy_one_does_not():
    rng = np.random.default_rng(1)
    M, K, N = 64, 512, 48
    W = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    cases = {"forward": (np.tanh(rng.standard_normal((M, K))).astype(np.float32), 1.0),
             "gradient": ((rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K))) / 32768).astype(np.float32), grad_scale(32768))}
    for name, (A, sa) in cases.items():
        ref = A.astype(np.float64) @ W.astype(np.float64)
        rms = lambda x: float(np.sqrt(np.mean((x.astype(np.float64) - ref) ** 2)))
        seq = _seq_fp32(A, W)
        three = _dot_products(A, W, [(0, 1), (1, 0), (0, 0)], sa, W_SCALE)
        one = _dot_products(A, W, [(0, 0)], sa, W_SCALE)
        assert rms(three) <= 1.0 * rms(seq) + 1e-15, (name, rms(three), rms(seq))
        assert rms(one) > 50.0 * rms(seq), name
        assert np.max(np.abs(three.astype(np.float64) - ref)) <= 4e-6 * np.sqrt(np.mean(ref ** 2)), name
