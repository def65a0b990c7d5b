"""GPU: the three MFMA GEMM kernels in isolation vs float64 numpy (asymmetric operands so a transposed fragment layout
cannot pass), in both engines: exact fp32 (v_mfma_f32_32x32x2_f32; dbg modes 0-2) and split-fp32 operands on the half-precision
matrix pipe (two fp16 planes per operand, three v_mfma_f32_32x32x16_f16 per 16 k, gemm_bx.h; dbg modes 3-5).  The split engine is
held to the SAME tolerances, and test_split_engine_error_budget additionally requires its fp64-referenced error to stay within
1.5x of the exact engine's.  Gradient operands (dZ) carry the pass's power-of-two scale in production (bx_grad_scale); the test
hook takes it from the option "bx_gscale_log2", set here so that a typical |dZ| lands near 8 like a per-sample gradient of 1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


class _gscale:
    """sets the split engine's gradient-operand scale for the calls inside (2^k with rms(dZ) * 2^k ~ 8), restores 1 after"""
    def __init__(self, ctx, dZ):
        self.ctx, self.k = ctx, int(np.clip(np.round(np.log2(8.0 / max(float(np.sqrt(np.mean(np.square(dZ, dtype=np.float64)))), 1e-30))), 0, 40))

    def __enter__(self):
        self.ctx.set_option("bx_gscale_log2", self.k)

    def __exit__(self, *a):
        self.ctx.set_option("bx_gscale_log2", 0)


def _elu_grad_from_out(h):
    return np.where(h > 0, 1.0, h + 1.0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 256, 512), (130, 132, 36), (1, 4, 4), (4096, 128, 256),
                                   (1000, 200, 100), (32805, 256, 512)])      # last: the 128-row / wave-specialised form + a ragged tile
@pytest.mark.parametrize("bx", [0, 3], ids=["fp32", "f16x3"])
def test_gemm_fwd(ctx, dev, M, N, K, bx):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32) * 0.1
    b = rng.standard_normal(N).astype(np.float32)
    C = torch.empty(M, N, device=dev)
    ctx.dbg_gemm(0 + bx, _t(A, dev), _t(W, dev), C, _t(b, dev), M, N, K, 1)
    z = A.astype(np.float64) @ W.astype(np.float64) + b
    exp = np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    np.testing.assert_allclose(C.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (256, 128, 256), (130, 36, 132), (4096, 256, 512), (777, 100, 60),
                                   (32805, 128, 256)])
@pytest.mark.parametrize("bx", [0, 3], ids=["fp32", "f16x3"])
def test_gemm_dx(ctx, dev, M, N, K, bx):
    rng = np.random.default_rng(M + N + K + 1)
    dZ = rng.standard_normal((M, N)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32) * 0.1
    H = (rng.standard_normal((M, K)) * 0.8).astype(np.float32)
    HD = _t(H, dev)
    with _gscale(ctx, dZ):
        ctx.dbg_gemm(1 + bx, _t(dZ, dev), _t(W, dev), HD, None, M, N, K, 1)
    exp = (dZ.astype(np.float64) @ W.astype(np.float64).T) * _elu_grad_from_out(H.astype(np.float64))
    np.testing.assert_allclose(HD.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)
    HD2 = _t(H, dev)
    with _gscale(ctx, dZ):
        ctx.dbg_gemm(1 + bx, _t(dZ, dev), _t(W, dev), HD2, None, M, N, K, -1)
    np.testing.assert_allclose(HD2.cpu().numpy(), dZ.astype(np.float64) @ W.astype(np.float64).T, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (4096, 256, 512), (32768, 128, 256), (1000, 132, 36), (31, 4, 8)])
@pytest.mark.parametrize("bx", [0, 3], ids=["fp32", "f16x3"])
def test_gemm_dw(ctx, dev, M, N, K, bx):
    rng = np.random.default_rng(M + N + K + 2)
    Hp = rng.standard_normal((M, K)).astype(np.float32)
    dZ = (rng.standard_normal((M, N)) / np.sqrt(M)).astype(np.float32)
    C = torch.empty(K, N, device=dev)
    db = torch.empty(N, device=dev)
    with _gscale(ctx, dZ):
        ctx.dbg_gemm(2 + bx, _t(Hp, dev), _t(dZ, dev), C, db, M, N, K, 0)
    exp = Hp.astype(np.float64).T @ dZ.astype(np.float64)
    # contraction over M rows of products of O(1) x O(1/sqrt(M)) numbers: the natural scale of an entry's error is the
    # Cauchy-Schwarz bound |h_col| |dz_col| (an entry itself may cancel to ~0), held to 1e-5 of it
    cs = np.sqrt((Hp.astype(np.float64) ** 2).sum(0))[:, None] * np.sqrt((dZ.astype(np.float64) ** 2).sum(0))[None, :]
    got = C.cpu().numpy().astype(np.float64)
    assert (np.abs(got - exp) / cs).max() < 1e-5
    assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 1e-5
    np.testing.assert_allclose(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5 * np.abs(dZ).sum(0).max())


def test_split_engine_error_budget(ctx, dev):
    """Same operands through both engines at the update's layer-2 shape; errors against float64.  Rows scaled over 3.5 decades
    (0.018 .. 55): inside the activation window of the split engine (|h| < 4094, full precision from 0.0078 up, gemm_bx.h)."""
    M, N, K = 8192, 256, 512
    rng = np.random.default_rng(99)
    A = (np.tanh(rng.standard_normal((M, K))) * np.exp(rng.uniform(-4, 4, (M, 1)))).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    b = np.zeros(N, np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.sqrt((A.astype(np.float64) ** 2).sum(1, keepdims=True)) * np.sqrt((W.astype(np.float64) ** 2).sum(0, keepdims=True))
    err = {}
    for mode in (0, 3):
        C = torch.empty(M, N, device=dev)
        ctx.dbg_gemm(mode, _t(A, dev), _t(W, dev), C, _t(b, dev), M, N, K, -1)
        e = np.abs(C.cpu().numpy().astype(np.float64) - ref) / scale
        err[mode] = (e.max(), np.sqrt((e ** 2).mean()))
    assert err[0][0] < 3e-7 and err[3][0] < 3e-7, err          # |error| relative to |a_row| |w_col| (Cauchy-Schwarz scale)
    assert err[3][1] <= 1.5 * err[0][1] + 1e-9, err
    # weight gradient: contraction over 32768 rows
    M2, N2, K2 = 32768, 128, 256
    Hp = np.tanh(rng.standard_normal((M2, K2))).astype(np.float32)
    dZ = (rng.standard_normal((M2, N2)) * np.exp(rng.uniform(-6, 0, (M2, 1))) / M2).astype(np.float32)
    refw = Hp.astype(np.float64).T @ dZ.astype(np.float64)
    errw = {}
    for mode in (2, 5):
        C = torch.empty(K2, N2, device=dev)
        db = torch.empty(N2, device=dev)
        with _gscale(ctx, dZ):
            ctx.dbg_gemm(mode, _t(Hp, dev), _t(dZ, dev), C, db, M2, N2, K2, 0)
        e = np.abs(C.cpu().numpy().astype(np.float64) - refw)
        errw[mode] = np.sqrt((e ** 2).mean()) / np.abs(refw).mean()
    assert errw[5] <= 1.5 * errw[2] + 1e-9 and errw[5] < 1e-5, errw
