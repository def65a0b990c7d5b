"""GPU: the plane-tensor GEMM kernels (rl-x_amd/csrc/gemm_px.hip) in isolation vs float64 numpy -- operands as two fp16 planes moved by
global_load_lds, weight-gradient fragments by ds_read_b64_tr_b16 -- through the test hook that converts fp32 buffers to planes
and back.  Asymmetric operands (a transposed fragment layout cannot pass), ragged row counts, both block-tile forms (N = 256:
one workgroup covers the full width; N = 128), the layer shapes of the bench (512 -> 256 -> 128)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _elu(z):
    return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))


def _gs_log2(dZ):
    return int(np.clip(np.round(np.log2(8.0 / max(float(np.sqrt(np.mean(np.square(dZ, dtype=np.float64)))), 1e-30))), 0, 40))


@pytest.mark.parametrize("M,N,K", [(4096, 256, 512), (4096, 128, 256), (32768, 256, 512), (32768, 128, 256), (5000, 256, 64),
                                   (4100, 128, 128), (8192, 512, 256)])
def test_forward_on_planes(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = np.tanh(rng.standard_normal((M, K))).astype(np.float32) * 2.0
    W = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    C = torch.full((M, N), float("nan"), device=dev)
    ctx.dbg_gemm_px(0, _t(A, dev), _t(W, dev), C, _t(b, dev), M, N, K, 1)
    exp = _elu(A.astype(np.float64) @ W.astype(np.float64) + b)
    got = C.cpu().numpy()
    assert np.isfinite(got).all()
    # the result itself is emitted as planes (22 significant bits): 2.4e-7 relative on top of the GEMM's own error
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=4e-6)


@pytest.mark.parametrize("M,N,K", [(4096, 128, 256), (32768, 128, 256), (4100, 64, 128), (8192, 256, 512)])
def test_input_gradient_on_planes(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K + 1)
    dZ = (rng.standard_normal((M, N)) * np.exp(rng.uniform(-3, 0, (M, 1))) / M).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    H = np.where(rng.random((M, K)) < 0.5, rng.standard_normal((M, K)), np.expm1(-np.abs(rng.standard_normal((M, K))))).astype(np.float32)
    HD = _t(H, dev)
    ctx.set_option("bx_gscale_log2", _gs_log2(dZ))
    try:
        ctx.dbg_gemm_px(1, _t(dZ, dev), _t(W, dev), HD, None, M, N, K, 1)
    finally:
        ctx.set_option("bx_gscale_log2", 0)
    # act'(H) from the planes of H: H itself carries 22 bits
    exp = (dZ.astype(np.float64) @ W.astype(np.float64).T) * np.where(H > 0, 1.0, H.astype(np.float64) + 1.0)
    got = HD.cpu().numpy()
    scale = np.abs(exp).max()
    assert np.isfinite(got).all()
    assert np.abs(got - exp).max() <= 2e-6 * scale
    assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 1e-6


@pytest.mark.parametrize("M,N,K", [(4096, 128, 256), (32768, 256, 512), (32768, 128, 256), (5001, 128, 128), (4097, 256, 128)])
def test_weight_gradient_on_planes(ctx, dev, M, N, K):
    rng = np.random.default_rng(M + N + K + 2)
    Hp = np.where(rng.random((M, K)) < 0.5, rng.standard_normal((M, K)), np.expm1(-np.abs(rng.standard_normal((M, K))))).astype(np.float32)
    dZ = (rng.standard_normal((M, N)) * np.exp(rng.uniform(-3, 0, (M, 1))) / M).astype(np.float32)
    C, db = torch.full((K, N), float("nan"), device=dev), torch.full((N,), float("nan"), device=dev)
    ctx.set_option("bx_gscale_log2", _gs_log2(dZ))
    try:
        ctx.dbg_gemm_px(2, _t(Hp, dev), _t(dZ, dev), C, db, M, N, K, 0)
    finally:
        ctx.set_option("bx_gscale_log2", 0)
    exp = Hp.astype(np.float64).T @ dZ.astype(np.float64)
    cs = np.sqrt((Hp.astype(np.float64) ** 2).sum(0))[:, None] * np.sqrt((dZ.astype(np.float64) ** 2).sum(0))[None, :]
    got = C.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    assert (np.abs(got - exp) / cs).max() < 1e-5
    assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 1e-5
    np.testing.assert_allclose(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5 * np.abs(dZ).sum(0).max())
