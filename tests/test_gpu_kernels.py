"""Kernel-level parity: each HIP kernel, called through the C ABI (revo_op_*),
against a plain PyTorch fp32 reference of the same op on the same inputs."""
import math
import os

import numpy as np

import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import _lib

pytestmark = pytest.mark.gpu

EPI_BF16, EPI_BF16_GELU, EPI_RESID_F32, EPI_F32 = 0, 1, 2, 3


@pytest.fixture(params=[0, 128, 256], autouse=True)
def gemm_tile(request):
    """Every test in this file runs once per GEMM tile configuration: 0 = the product library (librevo.so) with its
    size heuristic in charge; 128 / 256 = that tile forced.  The force / variant switches are not part of the product
    ABI: they exist only in librevo_exp.so (`make exp`, the same sources with -DREVO_EXPERIMENTS), which the forced
    runs load instead."""
    return request.param


@pytest.fixture
def lib(gemm_tile):
    if gemm_tile == 0:
        yield _lib.load()
        return
    exp = _lib.load_exp()
    _lib.check(exp.revo_op_set_gemm_tile(gemm_tile))
    yield exp
    _lib.check(exp.revo_op_set_gemm_tile(0))
    _lib.check(exp.revo_op_set_variant(0))


def _gemm(lib, epi, a, b, c, bias=None, gamma=None):
    M, K = a.shape
    N = b.shape[0]
    _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), M, N, K, _lib.ptr(c),
                                c.stride(0), _lib.ptr(bias), _lib.ptr(gamma), _lib.current_stream()), "gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1024), (300, 260, 128), (1154, 384, 128),
                                   (37, 1024, 1024), (1, 128, 64), (577, 3072, 1024), (129, 132, 4096),
                                   (2308, 1024, 1024), (1030, 516, 192), (256, 256, 64), (257, 260, 128)])
def test_gemm_f32(lib, dev, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = torch.randn(N, K, generator=g).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    c = torch.full((M, N), float("nan"), device=dev)
    _gemm(lib, EPI_F32, a, b, c, bias)
    ref = a.float() @ b.float().T + bias
    err = (c - ref).abs().max().item()
    assert err <= 2e-3 * math.sqrt(K / 64), err


def test_gemm_asymmetric_identity(lib, dev):
    # A = I, asymmetric B: catches a transposed C write or a swapped fragment map
    K = 128
    a = torch.eye(K, device=dev).bfloat16()
    b = (torch.arange(256 * K, device=dev).reshape(256, K) % 251).float().bfloat16()
    c = torch.zeros(K, 256, device=dev)
    _gemm(lib, EPI_F32, a, b, c)
    assert torch.equal(c, b.float().T)


def test_gemm_epilogues(lib, dev):
    M, N, K = 333, 512, 256
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    gamma = torch.randn(N, generator=g).to(dev)
    ref = a.float() @ b.float().T + bias
    # bf16 + bias
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _gemm(lib, EPI_BF16, a, b, c, bias)
    assert (c.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item()
    assert torch.equal(c, ref.bfloat16()) or (c.float() - ref.bfloat16().float()).abs().max().item() <= 0.07
    # exact-erf gelu
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _gemm(lib, EPI_BF16_GELU, a, b, c, bias)
    gref = torch.nn.functional.gelu(ref)
    assert (c.float() - gref).abs().max().item() <= 0.01 * max(1.0, gref.abs().max().item())
    # residual with layer scale, in place on fp32
    x0 = torch.randn(M, N, generator=g).to(dev)
    x = x0.clone()
    _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
    rref = x0 + gamma * ref
    assert (x - rref).abs().max().item() <= 2e-3
    x = x0.clone()
    _gemm(lib, EPI_RESID_F32, a, b, x, bias, None)
    assert (x - (x0 + ref)).abs().max().item() <= 2e-3


def test_gemm_rejects_bad_k(lib, dev):
    a = torch.zeros(8, 48, device=dev).bfloat16()
    b = torch.zeros(8, 48, device=dev).bfloat16()
    c = torch.zeros(8, 8, device=dev)
    rc = lib.revo_op_gemm(EPI_F32, _lib.ptr(a), 48, _lib.ptr(b), 48, 8, 8, 48, _lib.ptr(c), 8, None, None,
                          _lib.current_stream())
    assert rc != 0 and b"multiple of 64" in lib.revo_last_error()


@pytest.mark.parametrize("W", [128, 192, 768, 1024, 1536])
@pytest.mark.parametrize("out_bf16", [0, 1])
def test_layernorm(lib, dev, W, out_bf16):
    rows = 77
    g = torch.Generator(device="cpu").manual_seed(W)
    x = (torch.randn(rows, W, generator=g) * 3 + 1).to(dev)
    w = torch.randn(W, generator=g).to(dev)
    b = torch.randn(W, generator=g).to(dev)
    out = torch.zeros(rows, W, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _lib.check(lib.revo_op_layernorm(_lib.ptr(x), W, _lib.ptr(w), _lib.ptr(b), 1e-5, rows, W, _lib.ptr(out), W,
                                     out_bf16, _lib.current_stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x, (W,), w, b, 1e-5)
    tol = 0.04 if out_bf16 else 2e-5
    assert (out.float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("B,S,W,H", [(64, 577, 1024, 8), (48, 197, 768, 8), (5, 577, 1024, 8), (32, 1025, 1536, 8), (200, 17, 192, 3)])
def test_pool_rows_equal_the_softmax_weighted_sum_and_do_not_depend_on_the_batch(lib, dev, B, S, W, H):
    """K10's weighted row sums (head.hip pool_accumulate_kernel) against fp64 softmax-weighted sums, in both forms -- a batch
    (four columns per lane) and a few images (one column per lane): the same column sums, bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + S)
    x = torch.randn(B * S, W, generator=g).to(dev)
    logits = (torch.randn(B, H, S, generator=g) * 2).to(dev)
    u = torch.zeros(B, H, W, device=dev)
    _lib.check(lib.revo_op_pool_rows(_lib.ptr(x), W, _lib.ptr(logits), B, S, W, H, _lib.ptr(u), _lib.current_stream()))
    torch.cuda.synchronize()
    p = torch.softmax(logits.double(), dim=-1)
    ref = torch.einsum("bhs,bsw->bhw", p, x.double().view(B, S, W))
    assert (u.double() - ref).abs().max().item() <= 2e-5
    # image by image: the narrow form (few workgroups per image)
    for b in (0, B - 1):
        u1 = torch.zeros(1, H, W, device=dev)
        _lib.check(lib.revo_op_pool_rows(_lib.ptr(x[b * S:(b + 1) * S]), W, _lib.ptr(logits[b:b + 1].contiguous()), 1, S, W, H,
                                         _lib.ptr(u1), _lib.current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(u1[0], u[b]), b


@pytest.mark.parametrize("B,S,W,H", [(8, 577, 1024, 8), (24, 197, 768, 8), (5, 1025, 1536, 8), (3, 577, 1024, 8)])
def test_layernorm_with_pool_logits_batch_form_equals_the_row_form(lib, dev, B, S, W, H):
    """ln_post + the pool's logits (elementwise.hip): against torch, and the batch form (four rows per wave, >= 4096 rows)
    against the one-row-per-wave form image by image -- the same bits for rows and logits."""
    g = torch.Generator(device="cpu").manual_seed(B * 100 + S)
    rows = B * S
    x = (torch.randn(rows, W, generator=g) * 2 + 0.5).to(dev)
    w, b = torch.randn(W, generator=g).to(dev), torch.randn(W, generator=g).to(dev)
    qk, ck = (torch.randn(H, W, generator=g) * 0.05).to(dev), torch.randn(H, generator=g).to(dev)
    out, lg = torch.zeros(rows, W, device=dev), torch.zeros(B, H, S, device=dev)
    _lib.check(lib.revo_op_layernorm_logits(_lib.ptr(x), W, _lib.ptr(w), _lib.ptr(b), 1e-5, rows, W, _lib.ptr(out), W, _lib.ptr(qk),
                                            _lib.ptr(ck), H, S, _lib.ptr(lg), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (W,), w.double(), b.double(), 1e-5)
    assert (out.double() - ref).abs().max().item() <= 2e-5
    ref_lg = (ref @ qk.double().T + ck.double()).view(B, S, H).permute(0, 2, 1)
    assert (lg.double() - ref_lg).abs().max().item() <= 2e-4
    for i in (0, B - 1):
        o1, l1 = torch.zeros(S, W, device=dev), torch.zeros(1, H, S, device=dev)
        _lib.check(lib.revo_op_layernorm_logits(_lib.ptr(x[i * S:(i + 1) * S]), W, _lib.ptr(w), _lib.ptr(b), 1e-5, S, W, _lib.ptr(o1), W,
                                                _lib.ptr(qk), _lib.ptr(ck), H, S, _lib.ptr(l1), _lib.current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(o1, out[i * S:(i + 1) * S]) and torch.equal(l1[0], lg[i]), i


@pytest.mark.parametrize("M,N,K", [(64, 1024, 4096), (64, 4096, 1024), (50, 1024, 1024), (33, 1280, 1536), (7, 1024, 1024), (64, 1000, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_head_linear_f32_rows_do_not_depend_on_the_batch(lib, dev, M, N, K, epi):
    """The head's fp32 linear layers (head.hip): against fp64, and a batch (64-row passes, or 16 rows per workgroup where 16
    columns per workgroup would leave CUs idle) against its rows one at a time -- the same bits."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + epi)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    c0 = torch.randn(M, N, generator=g).to(dev)
    c = c0.clone()
    st = _lib.current_stream()
    _lib.check(lib.revo_op_linear_f32(epi, _lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(bias), M, N, K, _lib.ptr(c), N, st))
    torch.cuda.synchronize()
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + c0.double()
    assert (c.double() - ref).abs().max().item() <= 1e-4 * math.sqrt(K / 1024)
    for r in (0, M // 2, M - 1):
        c1 = c0[r:r + 1].clone()
        _lib.check(lib.revo_op_linear_f32(epi, _lib.ptr(a[r:r + 1]), K, _lib.ptr(w), K, _lib.ptr(bias), 1, N, K, _lib.ptr(c1), N, st))
        torch.cuda.synchronize()
        assert torch.equal(c1[0], c[r]), r


def test_layernorm_constant_row_is_bias(lib, dev):
    W = 1024
    x = torch.full((4, W), 3.25, device=dev)
    w = torch.randn(W, device=dev)
    b = torch.randn(W, device=dev)
    out = torch.zeros(4, W, device=dev)
    _lib.check(lib.revo_op_layernorm(_lib.ptr(x), W, _lib.ptr(w), _lib.ptr(b), 1e-5, 4, W, _lib.ptr(out), W, 0,
                                     _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, b.expand(4, W))


def _rope_table(S, hd, g, cls, theta=10000.0):
    from oracle import pe_vit

    class C:
        pass
    c = C()
    c.width, c.heads, c.image_size, c.patch_size, c.use_cls, c.rope_theta = hd, 1, g, 1, cls, theta
    ang = pe_vit.rope_angles(c, torch.float64)          # [S, hd], pairs repeated
    return ang


def test_rope(lib, dev):
    from oracle import pe_vit
    B, g, H, hd = 2, 5, 3, 64
    S, W = g * g + 1, 3 * 64
    ang = _rope_table(S, hd, g, True)
    cs = torch.stack([ang[:, 0::2].cos(), ang[:, 0::2].sin()], dim=-1).float().to(dev).contiguous()   # [S, hd/2, 2]
    gen = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn(B * S, 3 * W, generator=gen).to(dev).bfloat16()
    ref_in = qkv.float().cpu().double().reshape(B, S, 3, H, hd)
    q = pe_vit.apply_rope(ref_in[:, :, 0].transpose(1, 2), ang).transpose(1, 2)
    k = pe_vit.apply_rope(ref_in[:, :, 1].transpose(1, 2), ang).transpose(1, 2)
    _lib.check(lib.revo_op_rope(_lib.ptr(qkv), 3 * W, _lib.ptr(cs), B * S, S, W, H, _lib.current_stream()))
    torch.cuda.synchronize()
    got = qkv.float().cpu().double().reshape(B, S, 3, H, hd)
    assert (got[:, :, 0] - q).abs().max().item() <= 0.03
    assert (got[:, :, 1] - k).abs().max().item() <= 0.03
    assert torch.equal(got[:, :, 2], ref_in[:, :, 2])                       # v untouched
    assert torch.equal(got[:, 0, :2], ref_in[:, 0, :2])                     # cls token: identity rotation


@pytest.mark.parametrize("hd", [64, 96])
@pytest.mark.parametrize("B,S,H", [(2, 577, 2), (1, 197, 3), (3, 17, 2), (1, 64, 1), (1, 65, 1), (2, 128, 2),
                                   (1, 129, 1), (1, 1, 1), (1, 1024, 2), (2, 16, 2)])
def test_attention(lib, dev, B, S, H, hd):
    W = H * hd
    g = torch.Generator(device="cpu").manual_seed(S * 31 + H)
    qkv = torch.randn(B * S, 3 * W, generator=g).to(dev).bfloat16()
    out = torch.full((B * S, W), float("nan"), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, _lib.current_stream()))
    torch.cuda.synchronize()
    x = qkv.float().reshape(B, S, 3, H, hd)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * S, W)
    err = (out.float() - ref).abs().max().item()
    assert err <= 0.03, err


def test_attention_peaked_softmax(lib, dev):
    # one key dominates per query (forces the running max to jump late in the key sweep)
    B, S, H, hd = 1, 577, 1, 64
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(B * S, 3, hd, generator=g)
    x[:, 0] *= 0.1
    x[500, 1] = x[:, 0].mean(0) * 0 + 8.0 * torch.sign(x[3, 0])   # key 500 aligned with query 3
    qkv = x.reshape(B * S, 3 * hd).to(dev).bfloat16()
    out = torch.zeros(B * S, hd, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * hd, _lib.ptr(out), hd, B, S, H, hd, _lib.current_stream()))
    torch.cuda.synchronize()
    xf = qkv.float().reshape(S, 3, hd)
    att = torch.softmax(xf[:, 0] @ xf[:, 1].T * hd ** -0.5, dim=-1)
    ref = att @ xf[:, 2]
    assert (out.float() - ref).abs().max().item() <= 0.03


@pytest.mark.parametrize("S,hd,scale", [(577, 64, 8.0), (260, 64, 12.0), (257, 96, 8.0), (577, 64, 30.0)])
def test_attention_huge_logits(lib, dev, S, hd, scale):
    """Scores spread over hundreds of octaves (|s| up to ~scale^2 * 8): the optimistic softmax's
    reference must be moved (overflow guard) and rows whose early keys are far below the final
    maximum must still come out exact.  Compared with an fp64 softmax."""
    B, H = 2, 2
    W = H * hd
    g = torch.Generator(device="cpu").manual_seed(S + hd)
    qkv = torch.randn(B * S, 3 * W, generator=g)
    qkv[:, : 2 * W] *= scale
    # ascending scores along the key axis for the first head: the maximum keeps moving
    qkv[:, W:W + hd] += torch.linspace(0, scale, B * S)[:, None] * torch.sign(qkv[0, :hd])[None]
    qkv = qkv.to(dev).bfloat16()
    out = torch.full((B * S, W), float("nan"), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, _lib.current_stream()))
    torch.cuda.synchronize()
    x = qkv.double().reshape(B, S, 3, H, hd)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * S, W)
    assert torch.isfinite(out.float()).all()
    err = (out.double() - ref).abs().max().item()
    assert err <= 0.04, err


@pytest.mark.parametrize("B,S,H,scale", [(2, 577, 2, 0), (1, 197, 3, 0), (3, 17, 2, 0), (1, 64, 1, 0), (1, 65, 1, 0), (2, 128, 2, 0), (1, 129, 1, 0),
                                         (1, 1, 1, 0), (1, 1024, 2, 0), (2, 16, 2, 0), (64, 577, 16, 0),
                                         (2, 577, 2, 8.0), (2, 260, 2, 12.0), (2, 577, 2, 30.0)])
def test_attention_mfma_16x16x32_kernel(dev, gemm_tile, B, S, H, scale):
    """attn16_fwd_kernel (round 6's MFMA-shape experiment: head_dim 64 on v_mfma_f32_16x16x32_bf16 -- a query's scores in four
    lanes, probabilities fed back as the B operand in key-slot order, V rows swizzled for 4 x 16 transposing reads) against an fp64
    softmax, on every shape the 32x32x16 kernel is tested on (ragged last key tile, one row, class-token prelude, the rotated
    row order, logits over hundreds of octaves: the reference-moving rare path) and against that kernel itself."""
    if gemm_tile != 0:
        pytest.skip("runs once: the switch lives in librevo_exp.so")
    lib = _lib.load_exp()
    hd = 64
    W = H * hd
    g = torch.Generator(device="cpu").manual_seed(S * 31 + H)
    qkv = torch.randn(B * S, 3 * W, generator=g)
    if scale:
        qkv[:, : 2 * W] *= scale
        qkv[:, W:W + hd] += torch.linspace(0, scale, B * S)[:, None] * torch.sign(qkv[0, :hd])[None]
    qkv = qkv.to(dev).bfloat16()
    outs = {}
    try:
        for flag in (0, 1 << 20):
            _lib.check(lib.revo_op_set_variant(flag))
            out = torch.full((B * S, W), float("nan"), device=dev, dtype=torch.bfloat16)
            _lib.check(lib.revo_op_attention(_lib.ptr(qkv), 3 * W, _lib.ptr(out), W, B, S, H, hd, _lib.current_stream()))
            torch.cuda.synchronize()
            outs[flag] = out
    finally:
        _lib.check(lib.revo_op_set_variant(0))
    nb = min(B, 3)
    x = qkv[: nb * S].double().reshape(nb, S, 3, H, hd)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(nb * S, W)
    new, old = outs[1 << 20], outs[0]
    assert torch.isfinite(new.float()).all()
    tol = 0.04 if scale else 0.03
    assert (new[: nb * S].double() - ref).abs().max().item() <= tol
    assert (new.float() - old.float()).abs().max().item() <= tol           # the whole batch against the shipped kernel


@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (64, 4096, 1024), (64, 1024, 4096), (1, 256, 256), (7, 260, 512),
                                   (33, 1000, 768), (64, 1024, 1280)])
def test_gemm_skinny_all_epilogues(lib, dev, gemm_tile, M, N, K):
    """M <= 64 with the tile heuristic in charge (tile 0) goes to the skinny kernel: all four
    epilogues against an fp32 reference.  Runs once (not per forced tile)."""
    if gemm_tile != 0:
        pytest.skip("heuristic path: runs once, on the product library")
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    gamma = torch.rand(N, generator=g).to(dev) + 0.5
    ref = a.float() @ b.float().T + bias
    tol = 3e-3 * math.sqrt(K / 64)
    c = torch.full((M, N), float("nan"), device=dev)
    _gemm(lib, EPI_F32, a, b, c, bias)
    assert (c - ref).abs().max().item() <= tol
    cb = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    _gemm(lib, EPI_BF16, a, b, cb, bias)
    assert (cb.float() - ref).abs().max().item() <= tol + 0.02 * ref.abs().max().item()
    _gemm(lib, EPI_BF16_GELU, a, b, cb, bias)
    gref = torch.nn.functional.gelu(ref)
    assert (cb.float() - gref).abs().max().item() <= tol + 0.02 * gref.abs().max().item()
    x0 = torch.randn(M, N, generator=g).to(dev)
    x = x0.clone()
    _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
    assert (x - (x0 + gamma * ref)).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("M,N,K", [(4352, 4096, 128), (9000, 2304, 256), (4200, 4100, 64)])
def test_gemm_persistent_tile_loop(lib, dev, gemm_tile, M, N, K):
    """More than 256 output tiles: the 256x256 kernel runs its persistent form (one workgroup per CU
    walking several tiles, next tile's DMA issued under the epilogue).  Must equal the
    one-workgroup-per-tile form bit for bit, for every epilogue, and match an fp32 reference."""
    if gemm_tile != 256:
        pytest.skip("only the 256x256 kernel has a persistent form")
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.1).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    gamma = (torch.rand(N, generator=g) + 0.5).to(dev)
    x0 = torch.randn(M, N, generator=g).to(dev)
    ref = a.float() @ b.float().T + bias
    res = {}
    try:
        for flag in (1 << 16, 0):
            _lib.check(lib.revo_op_set_variant(flag))
            c32 = torch.full((M, N), float("nan"), device=dev)
            _gemm(lib, EPI_F32, a, b, c32, bias)
            cb = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            _gemm(lib, EPI_BF16, a, b, cb, bias)
            cg = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            _gemm(lib, EPI_BF16_GELU, a, b, cg, bias)
            x = x0.clone()
            _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
            res[flag] = (c32, cb, cg, x)
    finally:
        _lib.check(lib.revo_op_set_variant(0))
    for u, v in zip(res[1 << 16], res[0]):
        assert torch.equal(u, v)
    c32, cb, cg, x = res[0]
    tol = 3e-3 * math.sqrt(max(K, 64) / 64)
    assert (c32 - ref).abs().max().item() <= tol
    assert (cg.float() - torch.nn.functional.gelu(ref)).abs().max().item() <= tol + 0.02 * ref.abs().max().item()
    assert (x - (x0 + gamma * ref)).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("M,N,K", [(36928, 1024, 4096), (33000, 1024, 2048), (17000, 2048, 2048)])
def test_gemm_splitk_tail_residual(lib, dev, gemm_tile, M, N, K):
    """Residual GEMM whose last round of 256x256 tiles is partly empty: the leftover rows run with their K
    range cut across CUs (fp32 partial planes + a fixed-order reduce).  Same result as with the split
    disabled up to fp32 summation order, deterministic run to run, and right against an fp32 reference."""
    if gemm_tile != 128:
        pytest.skip("heuristic path with variant switches: one run on the experiment library is enough")
    _lib.check(lib.revo_op_set_gemm_tile(0))
    g = torch.Generator(device=dev).manual_seed(M + K)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    x0 = torch.randn(M, N, generator=g, device=dev)
    outs = []
    try:
        for flag in (0, 0, 1 << 17):
            _lib.check(lib.revo_op_set_variant(flag))
            x = x0.clone()
            _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
            outs.append(x)
    finally:
        _lib.check(lib.revo_op_set_variant(0))
    assert torch.equal(outs[0], outs[1])                                   # deterministic
    assert (outs[0] - outs[2]).abs().max().item() <= 1e-3                  # summation order only
    rows = torch.cat([torch.arange(0, 512, device=dev), torch.arange(M - 4200, M, device=dev)])
    ref = x0[rows] + gamma * (a[rows].float() @ b.float().T + bias)
    assert (outs[0][rows] - ref).abs().max().item() <= 3e-3 * math.sqrt(K / 64)


@pytest.mark.parametrize("M,N,K", [(36928, 1024, 1024), (36928, 1024, 4096), (36864, 1024, 1024), (12288 + 32, 4096, 512),
                                   (49152 + 128, 1536, 1536), (12288 + 16, 4096, 128), (12288 + 48, 4096, 192)])
def test_gemm_192_row_tiles_residual(lib, dev, gemm_tile, M, N, K):
    """Residual GEMMs whose rows make whole rounds of 192-row tiles but not of 256-row ones (PE-L14 at batch 64: 192 tile
    rows x 4 = three rounds; the 64 rows left over make the last four tile rows 208 rows tall): same bits as the 256-row
    plan wherever that plan runs its 256 x 256 kernel (same MFMA order along K, same epilogue arithmetic),
    summation-order differences in the rows the 256-row plan gives to its tail kernels, deterministic, and right
    against fp32 on rows around every kind of edge (192-row tile edges, the first tall tile, the last rows)."""
    if gemm_tile != 128:
        pytest.skip("heuristic path with variant switches: one run on the experiment library is enough")
    _lib.check(lib.revo_op_set_gemm_tile(0))
    g = torch.Generator(device=dev).manual_seed(M + K + 1)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    x0 = torch.randn(M, N, generator=g, device=dev)
    outs = []
    try:
        for flag in (0, 0, 1 << 3):
            _lib.check(lib.revo_op_set_variant(flag))
            x = x0.clone()
            _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
            outs.append(x)
    finally:
        _lib.check(lib.revo_op_set_variant(0))
    assert torch.equal(outs[0], outs[1])                                   # deterministic
    tn = (N + 255) // 256
    tiles = ((M + 255) // 256) * tn
    same = (tiles // 256 * 256) // tn * 256 if tiles % 256 else M          # rows of the 256-row plan's whole rounds
    assert same >= 8192
    assert torch.equal(outs[0][:same], outs[2][:same])
    assert (outs[0] - outs[2]).abs().max().item() <= 1e-3                  # summation order only
    T, e = M // 192, (M % 192) // 16
    first_tall = (T - e) * 192
    edges = [0, 190, 382, 192 * 64 - 2, same - 3, first_tall - 3, first_tall + 190, first_tall + 205, first_tall + 208 + 205, M - 210]
    rows = torch.cat([torch.arange(max(x_, 0), min(x_ + 6, M), device=dev) for x_ in edges if x_ < M] + [torch.arange(M - 70, M, device=dev)]).unique()
    ref = x0[rows] + gamma * (a[rows].float() @ b.float().T + bias)
    assert (outs[0][rows] - ref).abs().max().item() <= 3e-3 * math.sqrt(K / 64)
    # every row was written exactly once: against the 256-row plan no row differs by more than summation order, and
    # the rows outside [0, M) of a padded buffer are untouched
    xp = torch.full((M + 256, N), 7.0, device=dev)
    xp[:M] = x0
    _gemm(lib, EPI_RESID_F32, a, b, xp[:M], bias, gamma)
    assert torch.equal(xp[:M], outs[0]) and bool((xp[M:] == 7.0).all())


@pytest.mark.parametrize("M,N,K", [(1000, 4096, 512), (900, 2368, 1024), (640, 3008, 576)])
def test_gemm_128x64_variant_all_epilogues(lib, dev, gemm_tile, M, N, K):
    """Between 128 and 384 tiles of 128 x 128 (about one per CU) the heuristic takes 128 x 64 tiles so that
    several workgroups share a CU.  All four epilogues against an fp32 reference."""
    if gemm_tile != 0:
        pytest.skip("heuristic path: runs once, on the product library")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    ref = a.float() @ b.float().T + bias
    tol = 3e-3 * math.sqrt(K / 64)
    c = torch.full((M, N), float("nan"), device=dev)
    _gemm(lib, EPI_F32, a, b, c, bias)
    assert (c - ref).abs().max().item() <= tol
    cb = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    _gemm(lib, EPI_BF16_GELU, a, b, cb, bias)
    gref = torch.nn.functional.gelu(ref)
    assert (cb.float() - gref).abs().max().item() <= tol + 0.02 * gref.abs().max().item()
    x0 = torch.randn(M, N, generator=g, device=dev)
    x = x0.clone()
    _gemm(lib, EPI_RESID_F32, a, b, x, bias, gamma)
    assert (x - (x0 + gamma * ref)).abs().max().item() <= 2 * tol


def test_gemm_random_shapes_through_the_heuristic(lib, dev, gemm_tile):
    """Seeded random problem sizes with the tile heuristic in charge: whatever kernel family a size lands in
    (skinny, 128 x 64, 128 x 128, 256 x 256 per tile or persistent, split-K) must agree with an fp32 reference."""
    if gemm_tile != 0:
        pytest.skip("heuristic path: runs once, on the product library")
    rng = np.random.default_rng(2025)
    for case in range(40):
        M = int(rng.choice([1, 3, 17, 64, 65, 127, 128, 300, 577, 1024, 1154, 2308, 4160, 9232]))
        M = max(1, M + int(rng.integers(-2, 3)))
        N = int(rng.choice([4, 64, 132, 256, 768, 1024, 1280, 3072])) + 4 * int(rng.integers(0, 3))
        K = 64 * int(rng.choice([1, 2, 4, 9, 16, 24, 48, 64]))
        epi = int(rng.integers(0, 4))
        g = torch.Generator(device=dev).manual_seed(case)
        a = torch.randn(M, K, generator=g, device=dev).bfloat16()
        b = (torch.randn(N, K, generator=g, device=dev) * 0.1).bfloat16()
        bias = torch.randn(N, generator=g, device=dev)
        gamma = torch.rand(N, generator=g, device=dev) + 0.5
        ref = a.float() @ b.float().T + bias
        tol = 3e-3 * math.sqrt(K / 64)
        if epi == EPI_F32:
            c = torch.full((M, N), float("nan"), device=dev)
            _gemm(lib, epi, a, b, c, bias)
            err = (c - ref).abs().max().item()
        elif epi == EPI_RESID_F32:
            x0 = torch.randn(M, N, generator=g, device=dev)
            c = x0.clone()
            _gemm(lib, epi, a, b, c, bias, gamma)
            err = (c - (x0 + gamma * ref)).abs().max().item() / 2
        else:
            want = torch.nn.functional.gelu(ref) if epi == EPI_BF16_GELU else ref
            c = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            _gemm(lib, epi, a, b, c, bias)
            err = (c.float() - want).abs().max().item() - 0.02 * want.abs().max().item()
        assert err <= tol, (case, M, N, K, epi, err)


def test_gemm_repeat_determinism_large_grids(lib, dev, gemm_tile):
    """Every kernel family, grids larger than one resident set of workgroups: the same call five times
    must give the same bits (no timing-dependent reads)."""
    if gemm_tile != 0:
        pytest.skip("heuristic path: runs once, on the product library")
    g = torch.Generator(device=dev).manual_seed(77)
    for (M, N, K) in [(1970, 2304, 768), (2364, 3072, 768), (2167, 768, 3072), (4160, 1024, 1024), (9000, 1024, 4096),
                      (36928, 1024, 1024), (64, 4096, 1024), (1000, 4096, 512)]:
        a = torch.randn(M, K, generator=g, device=dev).bfloat16()
        b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, generator=g, device=dev)
        x0 = torch.randn(M, N, generator=g, device=dev)
        for epi in (EPI_BF16, EPI_BF16_GELU, EPI_RESID_F32, EPI_F32):
            outs = []
            for _ in range(5):
                c = x0.clone() if epi in (EPI_RESID_F32, EPI_F32) else torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
                _gemm(lib, epi, a, b, c, bias)
                outs.append(c)
            assert all(torch.equal(o, outs[0]) for o in outs[1:]), (M, N, K, epi)


@pytest.mark.parametrize("M,K,epi,lnfold,with_bias", [(36928, 1024, 0, True, True), (36928, 1024, 1, True, True),      # qkv- / fc1-shaped
                                                     (33000, 128, 0, False, True), (33100, 320, 1, False, False),     # two K-tiles; five, no bias
                                                     (70000, 192, 1, True, True), (36928, 1024, 5, True, True)])      # three K-tiles; RoPE
def test_gemm_queued_stores_kernel_equals_the_drained_one(dev, gemm_tile, M, K, epi, lnfold, with_bias):
    """gemm256q_kernel (round 6: a tile's values go from the accumulators to 16-byte row chunks in registers -- two
    v_permlane16_swap per fragment pair instead of an LDS transpose -- and its stores stay in flight through the next tile's
    first K-tile) against gemm256p_kernel (stores transposed through LDS and drained before the next main loop): the same
    bits, for every bf16 epilogue, with and without the folded LayerNorm and the bias, with rows past the matrix edge in the
    last tile row (dropped by the store descriptor) and with two to sixteen K-tiles; and against fp32 torch."""
    if gemm_tile != 0:
        pytest.skip("runs once: the switch lives in librevo_exp.so")
    lib = _lib.load_exp()
    N = 2560 if K != 1024 else (3072 if epi != 1 else 4096)
    g = torch.Generator(device=dev).manual_seed(M + K + epi)
    x = torch.randn(M, K, generator=g, device=dev) * (torch.rand(M, 1, generator=g, device=dev) * 2 + 0.3) + torch.randn(M, 1, generator=g, device=dev)
    a = x.bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device=dev) if with_bias else None
    st = _lib.current_stream()
    csum = b.float().sum(1)
    stats = None
    if lnfold and K % 256 == 0:
        xs = x.view(M, K // 256, 256)
        mm = xs.mean(2)
        stats = torch.stack([mm, ((xs - mm[..., None]) ** 2).sum(2)], dim=-1).contiguous()
    S, hd = 577, 64
    cs = torch.randn(S, hd // 2, 2, generator=g, device=dev)

    def run():
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        if epi == 5:
            _lib.check(lib.revo_op_gemm_rope(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(cs), S, hd,
                                             2 * N // 3, st), "gemm_rope")
        elif stats is not None:
            _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(csum),
                                              _lib.ptr(stats), K // 256, 1e-5, None, st), "gemm_ln_in")
        else:
            _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), None, st), "gemm")
        torch.cuda.synchronize()
        return c
    try:
        _lib.check(lib.revo_op_set_qstores(0))
        drained = run()
        _lib.check(lib.revo_op_set_qstores(3 if epi == 5 else 1))      # (3: the RoPE form too -- built and bit-identical, not the default)
        queued = [run() for _ in range(3)]
    finally:
        _lib.check(lib.revo_op_set_qstores(1))
    assert torch.isfinite(queued[0].float()).all()
    assert all(torch.equal(q, queued[0]) for q in queued[1:])          # repeatable
    # rows the whole rounds of tiles leave over (fc1: 64) are done by the queued-stores launch itself, K split over eight waves:
    # another summation order than the skinny kernel's -- equal up to the last bit of a bf16 here and there, not bit for bit
    m_tiles = M - M % 256 if (K % 256 == 0 and epi != 5 and M % 256 and M % 256 <= 64) else M
    assert torch.equal(queued[0][:m_tiles], drained[:m_tiles])
    if m_tiles < M:
        d = (queued[0][m_tiles:].float() - drained[m_tiles:].float()).abs()
        assert d.max().item() <= 0.008 * drained[m_tiles:].float().abs().max().item() + 1e-3, d.max().item()
        assert (d > 0).float().mean().item() < 0.02
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(M - 300, M, device=dev)])
    ref = a[rows].double() @ b.double().T
    if stats is not None and epi != 5:
        xd = x[rows].double()
        mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
        ref = (ref - mean * csum.double()) / torch.sqrt(var + 1e-5)
    if bias is not None:
        ref = ref + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi != 5:
        err = (queued[0][rows].double() - ref).abs().max().item()
        assert err <= 0.006 * ref.abs().max().item() + 2e-3, err


@pytest.mark.parametrize("M,S,H,hd", [(577, 577, 4, 64), (1970, 197, 12, 64), (2364, 197, 12, 64), (4616, 577, 4, 64),
                                      (1024, 1024, 2, 96), (36928, 577, 2, 64), (130, 65, 2, 64),
                                      (36928, 577, 4, 64), (32768, 1024, 8, 96)])        # the last two: whole 256-column tiles (queued-stores kernel)
def test_gemm_with_fused_rope_equals_gemm_then_rope(lib, dev, gemm_tile, M, S, H, hd):
    """The QKV projection with RoPE in the epilogue (any tile family) == the plain projection followed by the
    stand-alone RoPE kernel, bit for bit, and is the same on every repeat."""
    W = H * hd
    N, K = 3 * W, 256
    g = torch.Generator(device=dev).manual_seed(M + hd)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    cs = torch.randn(S, hd // 2, 2, generator=g, device=dev)
    st = _lib.current_stream()
    outs = []
    for _ in range(4):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        _lib.check(lib.revo_op_gemm_rope(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(cs), S, hd,
                                         2 * W, st), "gemm_rope")
        torch.cuda.synchronize()
        outs.append(c)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    c0 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    _gemm(lib, EPI_BF16, a, b, c0, bias)
    _lib.check(lib.revo_op_rope(_lib.ptr(c0), N, _lib.ptr(cs), M, S, W, H, st))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], c0)


def test_merge_step_device_check(tmp_path):
    """The wave-level sort / merge / de-duplicate helpers of the scan's exact path (topk256.hip), compiled into a small
    driver and checked against std::set for 4000 random key sets with many duplicates (tests/native/merge_step_check.hip)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "merge_step_check")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "revers-o_amd", "csrc"),
                    os.path.join(root, "tests", "native", "merge_step_check.hip"), "-o", exe], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bad trials: 0 of 2000 (KSEL 64), 0 of 2000 (KSEL 32)" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
