"""CPU: the arithmetic identity behind the bf16-pipe GEMM engine (rl-x_amd/csrc/gemm_bx.h), restated in numpy.

  * an fp32 value splits into three bf16 planes a = a0 + a1 + a2 (round-to-nearest-even conversions of the exact fp32
    residuals) with |a - (a0 + a1 + a2)| <= 2^-24 |a| -- across the whole exponent range, not only near 1;
  * the six plane products with p + q <= 2, accumulated in fp32, reproduce a dot product with the same fp64-referenced error as
    a sequential fp32 fma chain; three products (p + q <= 1) do not -- which is why the kernels issue six MFMAs per 16 k.
The GPU suite checks the kernels themselves (tests/test_gpu_gemm.py); this file pins the numerical argument on the CPU."""
import numpy as np


def _bf16_rne(x):
    """float32 -> nearest-even bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 produces, widened again)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _split3(a):
    a = np.asarray(a, np.float32)
    a0 = _bf16_rne(a)
    r1 = (a - a0).astype(np.float32)          # exact in fp32
    a1 = _bf16_rne(r1)
    r2 = (r1 - a1).astype(np.float32)
    a2 = _bf16_rne(r2)
    return a0, a1, a2


def test_three_planes_carry_24_bits_over_the_whole_exponent_range():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-60, 60, 200000))).astype(np.float32)
    a0, a1, a2 = _split3(a)
    # residual subtractions are exact: recombining in float64 shows what the planes hold
    err = np.abs(a.astype(np.float64) - (a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64)))
    assert np.all(err <= 2.0 ** -24 * np.abs(a.astype(np.float64)))
    # every plane is a bf16 value (low 16 bits of the fp32 pattern are zero)
    for p in (a0, a1, a2):
        assert not np.any(p.view(np.uint32) & 0xFFFF)


def _dot_products(A, B, pairs):
    """sum over the listed plane pairs of A_p @ B_q, each product exact, accumulation in float32 per 16-k block like the MFMA
    chain (the exact order inside the matrix pipe is not specified; float32 block sums are the conservative model)."""
    Ap, Bp = _split3(A), _split3(B)
    M, K = A.shape
    out = np.zeros((M, B.shape[1]), np.float32)
    for k0 in range(0, K, 16):
        for p, q in pairs:
            out = (out + (Ap[p][:, k0:k0 + 16].astype(np.float64) @ Bp[q][k0:k0 + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return out


def test_six_products_match_fp32_accuracy_three_do_not():
    rng = np.random.default_rng(1)
    M, K, N = 64, 512, 48
    A = np.tanh(rng.standard_normal((M, K))).astype(np.float32)
    B = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    seq = np.zeros((M, N), np.float32)
    for k in range(K):                              # sequential fp32 fma chain = the exact-fp32 MFMA engine
        seq = (seq.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1].astype(np.float64)).astype(np.float32)
    six = _dot_products(A, B, [(1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0)])
    three = _dot_products(A, B, [(0, 1), (1, 0), (0, 0)])
    rms = lambda x: float(np.sqrt(np.mean((x.astype(np.float64) - ref) ** 2)))
    assert rms(six) <= 1.5 * rms(seq) + 1e-12
    assert rms(three) > 5.0 * rms(seq)
