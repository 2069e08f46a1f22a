"""LayerNorm folded into the GEMMs around it (oracle/pe_vit.py:146-155 <- core_system.py:341 is the
arithmetic being served: h = ln(x); qkv = h W^T + b  /  h = ln(x); mlp = gelu(h W^T + b)).

The gain and shift live in the consuming GEMM's weights (W' = bf16(gamma . W), b' = b + W beta); the residual GEMM that
writes a row also writes bf16(x) and the row's (mean, M2) per 256-column slice, and the consuming GEMM's epilogue applies
rstd * (acc - mean * csum) + b'.  Checked here, through the C ABI:

* producer: the fp32 residual result is bit-identical to the plain residual GEMM's; xb is exactly bf16 of it; the
  statistics merge to torch's mean / variance of the fp32 rows; launch forms that cannot fold say so (done = 0);
* consumer: every launch form (persistent and one-tile 256 x 256 kernels, 128-row kernels, the skinny kernel) against
  torch fp32 on the same operands, rows with a large common offset and outlier channels included;
* the whole forward: folded form against the LayerNorm-kernel form of the same library (experiment switch) and against
  the committed goldens (tests/test_gpu_l14_golden.py, test_gpu_g14_golden.py run with the fold in place)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import _lib, engine, weights

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)


def _resid_ln(lib, a, b, c, bias, gamma, want_fold=True):
    M, K = a.shape
    N = b.shape[0]
    xb = torch.full((M, N), float("nan"), device=a.device, dtype=torch.bfloat16)
    stats = torch.full((M, N // 256, 2), float("nan"), device=a.device)
    done = C.c_int32(-1)
    _lib.check(lib.revo_op_gemm_resid_ln(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias), _lib.ptr(gamma),
                                         _lib.ptr(xb), N, _lib.ptr(stats), C.byref(done), None, 0, 0, _lib.current_stream()), "gemm_resid_ln")
    torch.cuda.synchronize()
    return xb, stats, done.value


def _merged(stats, eps=1e-5):
    """(mean, rstd) per row from the [M][P] (mean, M2) slots, in fp64."""
    m, q = stats[..., 0].double(), stats[..., 1].double()
    P = stats.shape[1]
    mean = m.mean(1)
    var = (q.sum(1) + 256.0 * ((m - mean[:, None]) ** 2).sum(1)) / (256.0 * P)
    return mean, 1.0 / torch.sqrt(var + eps)


@pytest.mark.parametrize("M,N,K", [(36928, 1024, 1024), (36928, 1024, 4096), (32768, 1536, 1536), (73856, 1024, 1024)])
def test_residual_gemm_writes_bf16_rows_and_statistics(lib, dev, M, N, K):
    """PE-L14 at batch 64 (192-row tiles, four of them 208 rows tall), G14 at batch 32 (256-row tiles, six slots per row),
    and a 128-image batch."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    x0 = torch.randn(M, N, generator=g, device=dev) * 2 + torch.randn(M, 1, generator=g, device=dev) * 3      # rows with an offset
    x0[:, 7] *= 80.0                                                                                   # an outlier channel
    ref = x0.clone()
    _lib.check(lib.revo_op_gemm(2, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(ref), N, _lib.ptr(bias), _lib.ptr(gamma),
                                _lib.current_stream()), "gemm")
    x = x0.clone()
    xb, stats, done = _resid_ln(lib, a, b, x, bias, gamma)
    assert done == 1
    assert torch.equal(x, ref)                                   # the residual stream itself: the same bits as without the fold
    assert torch.equal(xb, x.bfloat16())                         # the next GEMM's A operand: bf16 of exactly those values
    assert torch.isfinite(stats).all()
    mean, rstd = _merged(stats)
    xd = x.double()
    tm, tv = xd.mean(1), xd.var(1, unbiased=False)
    assert ((mean - tm).abs() <= 1e-6 * (1 + tm.abs())).all(), float((mean - tm).abs().max())
    tr = 1.0 / torch.sqrt(tv + 1e-5)
    assert ((rstd - tr).abs() <= 2e-6 * tr).all(), float(((rstd - tr).abs() / tr).max())
    # run to run: the same bits (fixed slots, fixed merge order)
    x2 = x0.clone()
    xb2, stats2, _ = _resid_ln(lib, a, b, x2, bias, gamma)
    assert torch.equal(stats2, stats) and torch.equal(xb2, xb) and torch.equal(x2, x)


@pytest.mark.parametrize("M,N,K", [(577, 1024, 1024), (5000, 1024, 4096), (18464, 1024, 1024), (36928, 1000, 1024)])
def test_residual_gemm_forms_that_cannot_fold_say_so(lib, dev, M, N, K):
    """One image (split-K + reduce), a batch with leftover rows, a width that is not a multiple of 256: the residual
    result is the plain GEMM's and done = 0 -- the caller runs the LayerNorm kernel."""
    g = torch.Generator(device=dev).manual_seed(M + N)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    x0 = torch.randn(M, N, generator=g, device=dev)
    ref = x0.clone()
    _lib.check(lib.revo_op_gemm(2, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(ref), N, _lib.ptr(bias), None,
                                _lib.current_stream()), "gemm")
    x = x0.clone()
    M_, K_ = a.shape
    xb = torch.zeros((M, (N + 255) // 256 * 256), device=dev, dtype=torch.bfloat16)
    stats = torch.zeros((M, max(N // 256, 1), 2), device=dev)
    done = C.c_int32(-1)
    _lib.check(lib.revo_op_gemm_resid_ln(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(x), N, _lib.ptr(bias), None,
                                         _lib.ptr(xb), xb.stride(0), _lib.ptr(stats), C.byref(done), None, 0, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    assert done.value == 0 and torch.equal(x, ref)
    # ... and the stream in planes is refused by such a form, by name, instead of being half-honoured
    xlo = torch.zeros_like(xb)
    rc = lib.revo_op_gemm_resid_ln(_lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(x), N, _lib.ptr(bias), None, _lib.ptr(xb),
                                   xb.stride(0), _lib.ptr(stats), C.byref(done), _lib.ptr(xlo), 0, 1, _lib.current_stream())
    assert rc == -2 and b"planes" in lib.revo_last_error()


@pytest.mark.parametrize("M,N,K", [(36928, 1024, 1024), (36928, 1024, 4096), (32768, 1536, 8960)])
def test_residual_stream_in_two_bf16_planes(lib, dev, M, N, K):
    """Between folded GEMMs the forward keeps the stream as hi = bf16(x), lo = bf16(x - hi).  A chain of three residual
    GEMMs -- fp32 in / planes out, planes in / planes out, planes in / fp32 out -- against the same three on fp32 rows:
    every intermediate hi + lo equals the fp32 stream to 2^-16 of the value, hi is exactly bf16 of the stream it was
    split from, the statistics are those of the unsplit values, and the final fp32 rows agree to the same bound."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = [torch.randn(M, K, generator=g, device=dev).bfloat16() for _ in range(3)]
    b = [(torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16() for _ in range(3)]
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    x0 = torch.randn(M, N, generator=g, device=dev) * 2 + torch.randn(M, 1, generator=g, device=dev)
    x0[:, 9] *= 70.0
    st = _lib.current_stream()
    ref = x0.clone()
    refs = []
    for i in range(3):
        _lib.check(lib.revo_op_gemm(2, _lib.ptr(a[i]), K, _lib.ptr(b[i]), K, M, N, K, _lib.ptr(ref), N, _lib.ptr(bias), _lib.ptr(gamma), st))
        torch.cuda.synchronize()
        refs.append(ref.clone())
    x = x0.clone()
    hi = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    lo = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    stats = torch.full((M, N // 256, 2), float("nan"), device=dev)
    done = C.c_int32(-1)

    def call(i, stats_t, x_in_planes, planes_out):
        _lib.check(lib.revo_op_gemm_resid_ln(_lib.ptr(a[i]), K, _lib.ptr(b[i]), K, M, N, K, _lib.ptr(x), N, _lib.ptr(bias), _lib.ptr(gamma),
                                             _lib.ptr(hi), N, _lib.ptr(stats_t), C.byref(done), _lib.ptr(lo), x_in_planes, planes_out, st),
                   "gemm_resid_ln")
        torch.cuda.synchronize()

    call(0, stats, 0, 1)
    assert done.value == 1 and torch.equal(x, x0)                       # fp32 rows untouched: the stream moved to the planes
    y = hi.float() + lo.float()
    tol = lambda r: 2.0 ** -16 * r.abs() + 1e-30        # (each split rounds to 2^-17 of the value it splits)
    assert ((y - refs[0]).abs() <= tol(refs[0])).all()
    assert torch.equal(hi, y.bfloat16()) or (hi.float() - refs[0]).abs().max() <= 2.0 ** -8 * refs[0].abs().max()
    mean, rstd = _merged(stats)
    rd = refs[0].double()
    assert ((mean - rd.mean(1)).abs() <= 1e-6 * (1 + rd.mean(1).abs())).all()
    tr = 1.0 / torch.sqrt(rd.var(1, unbiased=False) + 1e-5)
    assert ((rstd - tr).abs() <= 2e-6 * tr).all()
    call(1, stats, 1, 1)
    y = hi.float() + lo.float()
    assert ((y - refs[1]).abs() <= tol(refs[0]) + tol(refs[1])).all()             # the earlier split's error stays, absolutely
    call(2, None, 1, 0)
    assert ((x - refs[2]).abs() <= tol(refs[0]) + tol(refs[1]) + tol(refs[2])).all()
    assert torch.isfinite(x).all()


def _consumer_case(dev, M, N, K, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(M, K, generator=g, device=dev) * (torch.rand(M, 1, generator=g, device=dev) * 4 + 0.2)
    x += torch.randn(M, 1, generator=g, device=dev) * 2.0                    # a per-row offset of the order of the spread
    x[:, 11] *= 60.0
    x[:, 500 % K] *= -40.0                                                    # outlier channels (trained towers have them)
    wq = (torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16()      # W' = bf16(gamma . W), as stored
    bias = torch.randn(N, generator=g, device=dev)
    csum = wq.float().sum(1)
    P = K // 256
    xs = x.view(M, P, 256)
    m = xs.mean(2)
    q = ((xs - m[..., None]) ** 2).sum(2)
    stats = torch.stack([m, q], dim=-1).contiguous()                          # the producer's format
    return x, wq, bias, csum, stats


@pytest.mark.parametrize("M,N,K,epi", [(36928, 3072, 1024, 0), (36928, 4096, 1024, 1),       # persistent kernel (+ the skinny leftover rows of fc1)
                                       (2304, 3072, 1024, 0),                                 # one-tile 256 x 256 kernel
                                       (1000, 1024, 1024, 1), (300, 512, 768, 0),             # 128-row kernels
                                       (64, 4096, 1024, 1), (33, 1024, 1536, 0),              # skinny kernel
                                       (32768, 4608, 1536, 0), (8192, 8960, 1536, 1)])        # G14 shapes (six slots; tail split)
def test_consuming_gemm_applies_the_row_statistics(lib, dev, M, N, K, epi):
    x, wq, bias, csum, stats = _consumer_case(dev, M, N, K, M + N + K + epi)
    xb = x.bfloat16()
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(xb), K, _lib.ptr(wq), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias),
                                      _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, None, _lib.current_stream()), "gemm_ln_in")
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, min(M, 400), device=dev), torch.arange(max(M - 400, 0), M, device=dev),
                      torch.arange(M // 2, min(M // 2 + 300, M), device=dev)]).unique()
    xd = x[rows].double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    # the kernel's arithmetic in fp64: the bf16-rounded row against the rounded weights, the exact mean and rstd
    ref = ((xb[rows].double() @ wq.double().T) - mean * csum.double()) * rstd + bias.double()
    # ... and the LayerNorm it stands for (fp64, unrounded row): they differ by the rounding of x alone
    ln = (((xd - mean) * rstd) @ wq.double().T) + bias.double()
    if epi == 1:
        ref, ln = torch.nn.functional.gelu(ref), torch.nn.functional.gelu(ln)
    got = out[rows].double()
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 0.006 * scale + 2e-3, ((got - ref).abs().max().item(), scale)      # bf16 output rounding
    assert (got - ln).abs().max().item() <= 0.03 * scale, ((got - ln).abs().max().item(), scale)                 # + bf16 rounding of x
    # against the ordinary path on the same row: LayerNorm kernel (no affine) -> bf16 -> plain GEMM
    h = torch.empty((len(rows), K), device=dev, dtype=torch.bfloat16)
    xr = x[rows].contiguous()
    _lib.check(lib.revo_op_layernorm(_lib.ptr(xr), K, None, None, 1e-5, len(rows), K, _lib.ptr(h), K, 1, _lib.current_stream()))
    plain = torch.empty((len(rows), N), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_gemm(epi, _lib.ptr(h), K, _lib.ptr(wq), K, len(rows), N, K, _lib.ptr(plain), N, _lib.ptr(bias), None,
                                _lib.current_stream()))
    torch.cuda.synchronize()
    e_fold = (got - ln).pow(2).mean().sqrt().item()
    e_plain = (plain.double() - ln).pow(2).mean().sqrt().item()
    print(f"rms error against the fp64 LayerNorm + linear: folded {e_fold:.3e}, LayerNorm kernel + GEMM {e_plain:.3e}")
    assert e_fold <= 2.5 * e_plain + 1e-4, (e_fold, e_plain)      # rows with an offset of the order of their spread: same class of error


@pytest.mark.parametrize("M,N,epi", [(4096, 3072, 0), (4096, 3072, 1), (16384, 3072, 0)])     # one-tile 256 x 256 kernel; persistent (queued stores): whole rounds,
                                                                                             # no leftover rows on other kernels (those are not sampled)
def test_fold_error_against_the_row_offset(lib, dev, M, N, epi):
    """The regime the fold is weakest in: A = bf16(x) is NOT centred, so bf16 spends its 8 bits on a row's common offset.
    Rows with offsets of 0 / 2 / 8.5 / 33 standard deviations (each block of rows at one offset), folded consumer against
    LayerNorm kernel -> bf16 -> plain GEMM, both against the fp64 LayerNorm + linear: the ratio e_fold / e_plain per offset is
    printed (DESIGN.md quotes it) and held to a bound a trained checkpoint would be held to, and the consumer's telemetry
    (include/revo.h revo_vit_stats: rows with |mean| * rstd > 8) counts exactly the rows of the 8.5- and 33-sigma blocks."""
    K = 1024
    offs = [0.0, 2.0, 8.5, 33.0]                      # (8.5 / 33: safely on the far side of the telemetry's ratios 8 and 32)
    g = torch.Generator(device=dev).manual_seed(77 + epi)
    z = torch.randn(M, K, generator=g, device=dev)
    z = (z - z.mean(1, keepdim=True)) / z.std(1, unbiased=False, keepdim=True)           # rows with mean 0, std 1 exactly
    sigma = torch.rand(M, 1, generator=g, device=dev) * 3 + 0.3
    blk = M // len(offs)
    off = torch.cat([torch.full((blk, 1), o, device=dev) for o in offs])
    sign = torch.where(torch.rand(M, 1, generator=g, device=dev) < 0.5, -1.0, 1.0)
    x = sigma * (z + sign * off)
    wq = (torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    csum = wq.float().sum(1)
    xs = x.view(M, K // 256, 256)
    m = xs.mean(2)
    stats = torch.stack([m, ((xs - m[..., None]) ** 2).sum(2)], dim=-1).contiguous()
    xb = x.bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    tele = torch.zeros(4, dtype=torch.int64, device=dev)
    _lib.check(lib.revo_op_gemm_ln_in(epi, _lib.ptr(xb), K, _lib.ptr(wq), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias),
                                      _lib.ptr(csum), _lib.ptr(stats), K // 256, 1e-5, _lib.ptr(tele), _lib.current_stream()), "gemm_ln_in")
    h = torch.empty((M, K), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_layernorm(_lib.ptr(x), K, None, None, 1e-5, M, K, _lib.ptr(h), K, 1, _lib.current_stream()))
    plain = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_gemm(epi, _lib.ptr(h), K, _lib.ptr(wq), K, M, N, K, _lib.ptr(plain), N, _lib.ptr(bias), None,
                                _lib.current_stream()))
    torch.cuda.synchronize()
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    ln = (((xd - mean) / torch.sqrt(var + 1e-5)) @ wq.double().T) + bias.double()
    if epi == 1:
        ln = torch.nn.functional.gelu(ln)
    ratios = {}
    for j, o in enumerate(offs):
        r = slice(j * blk, (j + 1) * blk)
        e_fold = (out[r].double() - ln[r]).pow(2).mean().sqrt().item()
        e_plain = (plain[r].double() - ln[r]).pow(2).mean().sqrt().item()
        ratios[o] = e_fold / e_plain
        print(f"offset {o:5.1f} sigma: rms error folded {e_fold:.3e}, LayerNorm kernel + GEMM {e_plain:.3e}, ratio {ratios[o]:.2f}")
    # the bound: no worse than the ordinary path at no offset, and growing no faster than the offset itself
    # (measured: 1.00 / 1.51 / 4.88 / 18.7 -- about 0.57 x the offset in sigmas; include/revo.h revo_vit_stats quotes it)
    assert ratios[0.0] <= 1.1 and ratios[2.0] <= 2.0 and ratios[8.5] <= 6.5 and ratios[33.0] <= 24.0, ratios
    t = tele.cpu().tolist()
    assert t[0] == M and t[1] == 2 * blk and t[2] == blk, t        # every row once; > 8: the 8.5- and 33-sigma blocks; > 32: the last


def test_forward_with_the_fold_equals_the_layernorm_kernel_form(dev):
    """PE-L14 at batch 64 -- every ln_1 (from block 1 on) and ln_2 folded -- against the same library with the LayerNorm
    kernels in front of the same (gain-folded) weights, and the trained-tower regime (LayerNorm gains of 20, residual
    channels 270 x the median) likewise."""
    import make_golden_l14 as mg
    exp = _lib.load_exp()
    cfg, sd, u8 = mg.batch_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=64, experiments=True)
    img = u8.to(dev)
    eng.ln_fold_stats(reset=True)
    e_fold = eng.embed(img)
    # the fold's run-time telemetry (include/revo.h revo_vit_stats): every row of every consuming GEMM of the forward is counted
    # once -- ln_1 of blocks 1..23 (qkv: all 36 928 rows are tile rows) and ln_2 of all 24 (fc1: the 64 rows its tiles leave over
    # are finished, and sampled, by the launch's fused tail) -- and a random-init tower has no row beyond 8 sigma
    st = eng.ln_fold_stats(reset=True)
    assert st["rows"] == (23 + 24) * 64 * 577, st
    assert st["rows_above_ratio"] == 0 and st["rows_above_4x_ratio"] == 0 and st["ratio"] == 8.0, st
    assert eng.ln_fold_stats()["rows"] == 0
    x_fold = eng.residual_after(img[:2], 24)            # (two images: the LayerNorm-kernel path either way)
    assert exp.revo_debug_stream_in_planes(eng._h, 64) == 1 and exp.revo_debug_stream_in_planes(eng._h, 2) == 0
    _lib.check(exp.revo_op_set_ln_fold(0))
    try:
        e_kern = eng.embed(img)
        _lib.check(exp.revo_op_set_ln_fold(2))            # folded, but the stream as fp32 rows + a bf16 copy
        e_f32 = eng.embed(img)
    finally:
        _lib.check(exp.revo_op_set_ln_fold(1))
    cos = (e_fold * e_kern).sum(-1)
    assert float(cos.min()) >= 0.99999, float(cos.min())
    assert not torch.equal(e_fold, e_kern)                # the switch did switch forms
    # planes against fp32 rows under the same folded arithmetic: the stream's 2^-17 against bf16 operands' 2^-9
    assert float((e_fold * e_f32).sum(-1).min()) >= 0.999999 and not torch.equal(e_fold, e_f32)
    # a forward that stops inside the folded stretch hands back fp32 rows all the same (the last executed fc2 leaves the planes)
    x12 = eng.residual_after(img, 12)
    _lib.check(exp.revo_op_set_ln_fold(2))
    try:
        x12_f32 = eng.residual_after(img, 12)
    finally:
        _lib.check(exp.revo_op_set_ln_fold(1))
    # (two realisations of the bf16 rounding noise: a 2^-17 change of the stream flips one bf16 rounding of the GEMM operands
    #  in ~256, so they sit as far from each other as each sits from the oracle -- block 11: 3.9e-3, tests/test_gpu_l14_golden.py)
    assert float((x12 - x12_f32).norm() / x12_f32.norm()) <= 6e-3 and torch.isfinite(x12).all()
    # the oracle's per-block activations of two of the 64 images (tests/golden/l14_batch64.npz), taken INSIDE the batch:
    # the folded LayerNorms and the stream in planes land where the LayerNorm-kernel path of the two-image forwards lands
    # (tests/test_gpu_l14_golden.py: block 6 / 12 / 24 at 3.1 / 3.9 / 4.6e-3, asserted at 5 / 6 / 7e-3)
    gold = np.load(os.path.join(HERE, "golden", "l14_batch64.npz"))
    timg, ttok = gold["tap_images"].tolist(), gold["tap_tokens"].tolist()
    rel = {}
    for b, bound in zip(gold["tap_blocks"].tolist(), (5e-3, 6e-3, 7e-3)):
        x = eng.residual_after(img, b + 1)[timg][:, ttok].cpu()
        r = torch.from_numpy(gold[f"tap_block{b}"])
        rel[b] = float((x - r).norm() / r.norm())
        assert rel[b] <= bound, rel
    print("L14 batch 64, folded LayerNorm + planes: relative distance from the oracle after blocks", {k: round(v, 5) for k, v in rel.items()})
    assert torch.isfinite(x_fold).all()
    assert torch.equal(eng.embed(img), e_fold)            # repeatable bit for bit
    eng.close()
    cfg, sd, u8o, big = mg.outlier_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=64, experiments=True)
    img = u8o.to(dev).repeat(16, 1, 1, 1)                 # 64 images: the folded form
    eng.ln_fold_stats(reset=True)
    e_fold = eng.embed(img)
    st = eng.ln_fold_stats()
    print("outlier tower, rows beyond the telemetry's ratios:", st)
    assert st["rows"] > 0 and st["rows_above_4x_ratio"] <= st["rows_above_ratio"] <= st["rows"]
    _lib.check(exp.revo_op_set_ln_fold(0))
    try:
        e_kern = eng.embed(img)
    finally:
        _lib.check(exp.revo_op_set_ln_fold(1))
    assert float((e_fold * e_kern).sum(-1).min()) >= 0.99999
    gold = np.load(os.path.join(HERE, "golden", "l14_outlier.npz"))
    ref = torch.from_numpy(gold["embedding"]).repeat(16, 1)
    from _parity import assert_embeddings_match
    assert_embeddings_match(e_fold.cpu(), ref, what="outlier tower, folded LayerNorm at batch 64")
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(2000, cfg.out_dim, generator=g), dim=-1)
    assert ((e_fold.cpu() @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    eng.close()
