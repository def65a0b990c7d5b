"""GPU: the three exact-fp32 MFMA GEMM kernels in isolation vs float64 numpy (asymmetric
operands so a transposed fragment layout cannot pass)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _elu_grad_from_out(h):
    return np.where(h > 0, 1.0, h + 1.0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 256, 512), (130, 132, 36), (1, 4, 4), (4096, 128, 256),
                                   (1000, 200, 100)])
def test_gemm_fwd(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32) * 0.1
    b = rng.standard_normal(N).astype(np.float32)
    C = torch.empty(M, N, device=dev)
    ctx.dbg_gemm(0, _t(A, dev), _t(W, dev), C, _t(b, dev), M, N, K, 1)
    z = A.astype(np.float64) @ W.astype(np.float64) + b
    exp = np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    np.testing.assert_allclose(C.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (256, 128, 256), (130, 36, 132), (4096, 256, 512), (777, 100, 60)])
def test_gemm_dx(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K + 1)
    dZ = rng.standard_normal((M, N)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32) * 0.1
    H = (rng.standard_normal((M, K)) * 0.8).astype(np.float32)
    HD = _t(H, dev)
    ctx.dbg_gemm(1, _t(dZ, dev), _t(W, dev), HD, None, M, N, K, 1)
    exp = (dZ.astype(np.float64) @ W.astype(np.float64).T) * _elu_grad_from_out(H.astype(np.float64))
    np.testing.assert_allclose(HD.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)
    HD2 = _t(H, dev)
    ctx.dbg_gemm(1, _t(dZ, dev), _t(W, dev), HD2, None, M, N, K, -1)
    np.testing.assert_allclose(HD2.cpu().numpy(), dZ.astype(np.float64) @ W.astype(np.float64).T, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (4096, 256, 512), (32768, 128, 256), (1000, 132, 36), (31, 4, 8)])
def test_gemm_dw(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K + 2)
    Hp = rng.standard_normal((M, K)).astype(np.float32)
    dZ = (rng.standard_normal((M, N)) / np.sqrt(M)).astype(np.float32)
    C = torch.empty(K, N, device=dev)
    db = torch.empty(N, device=dev)
    ctx.dbg_gemm(2, _t(Hp, dev), _t(dZ, dev), C, db, M, N, K, 0)
    exp = Hp.astype(np.float64).T @ dZ.astype(np.float64)
    np.testing.assert_allclose(C.cpu().numpy(), exp, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=1e-4, atol=2e-5)
