"""BASELINE.json configs[4]'s tower as written: PE-Core-G14-448 at 32 images per GPU (256 images over 8 GPUs), full
depth (50 blocks, width 1536, 16 heads x 96, MLP 8960, 1024 tokens without a class token, pool MLP 6144, 1280-d output).

* the two golden images (tests/golden/make_golden_g14.py: the CPU oracle's embedding + per-block activations) INSIDE the
  batch of 32 against the golden vectors;
* batch invariance (the same image alone and in the batch: other GEMM tilings, same vector up to bf16 noise) and
  repeat determinism (split-K partials reduced in a fixed order);
* the three body-GEMM shapes of that batch -- (M, N, K) = (32768, 4608, 1536) qkv, (32768, 8960, 1536) fc1,
  (32768, 1536, 8960) fc2 (35 N-tiles with 4 stripes, 140 K-tiles, a 768-tile out-proj = exactly 3 rounds) -- against
  torch fp32 on sampled rows."""
import math
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import _lib, engine

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import make_golden_g14 as mg  # noqa: E402
from _parity import assert_embeddings_match  # noqa: E402


def test_g14_batch32_full_depth_vs_golden(dev):
    gold = np.load(os.path.join(HERE, "golden", "g14_batch32.npz"))
    cfg, sd, u8 = mg.batch_case()
    assert int(u8.long().sum()) == int(gold["image_sum"])
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=mg.BATCH)
    # the per-block taps come from a handle of librevo_exp.so (parity-test hooks; same sources as the product library)
    engx = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2, experiments=True)
    del sd
    imgs = u8.to(dev)
    emb = eng.embed(imgs)
    assert emb.shape == (mg.BATCH, 1280) and torch.isfinite(emb).all()
    assert ((emb.norm(dim=-1) - 1).abs() <= 1e-5).all()
    gi = gold["images"].tolist()
    ref = torch.from_numpy(gold["embedding"])
    cos = torch.nn.functional.cosine_similarity(emb[gi].cpu().double(), ref.double(), dim=-1)
    assert (cos >= 0.9999).all(), cos
    # the image-specific part: remove the batch's common direction (the oracle's own mean is not available for 32
    # images; the engine's batch mean serves both sides -- a shared direction cannot hide a per-image error)
    mean = emb.mean(0, keepdim=True).cpu()
    cc = torch.nn.functional.cosine_similarity((emb[gi].cpu() - mean).double(), (ref - mean).double(), dim=-1)
    assert (cc >= 0.99).all(), cc
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(2000, 1280, generator=g), dim=-1)
    assert ((emb[gi].cpu() @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    # repeat determinism
    assert torch.equal(eng.embed(imgs), emb)
    # batch invariance: the golden images alone (other tile shapes, split-K forms)
    alone = eng.embed(imgs[gi])
    assert (torch.nn.functional.cosine_similarity(alone, emb[gi], dim=-1) >= 0.99995).all()
    assert_embeddings_match(alone.cpu(), ref, centred_min=-1.0)
    # per-block activations of the two golden images (embedded as a batch of two)
    ttok = gold["tap_tokens"].tolist()
    two = imgs[gi]
    assert torch.equal(engx.embed(two), eng.embed(two))
    rel = {}
    for b in gold["tap_blocks"].tolist():
        x = engx.residual_after(two, b + 1)[:, ttok].cpu()
        r = torch.from_numpy(gold[f"tap_block{b}"])
        rel[f"block{b}"] = float((x - r).norm() / r.norm())
    taps = engx.taps(two)
    r = torch.from_numpy(gold["tap_ln_post"])
    rel["ln_post"] = float((taps["ln_post"][:, ttok].cpu() - r).norm() / r.norm())
    r = torch.from_numpy(gold["tap_pooled"])
    rel["pooled"] = float((taps["pooled"].cpu() - r).norm() / r.norm())
    rel["embedding"] = float((taps["embedding"].cpu() - ref).norm() / ref.norm())
    print("G14 relative distance from the oracle by stage:", {k: round(v, 5) for k, v in rel.items()})
    assert max(rel.values()) <= 1.2e-2 and rel["embedding"] <= 8e-3, rel
    eng.close()
    engx.close()


@pytest.mark.parametrize("M,N,K,epi", [(32768, 4608, 1536, 0), (32768, 8960, 1536, 1), (32768, 1536, 8960, 2),
                                       (32768, 1536, 1536, 2)])
def test_g14_batch32_gemm_shapes_vs_torch_fp32(lib, dev, M, N, K, epi):
    """The launcher branches configs[4] takes at 32 images per GPU, through the product library's own heuristics."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    b = (torch.randn(N, K, generator=g, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, generator=g, device=dev)
    gamma = torch.rand(N, generator=g, device=dev) + 0.5
    if epi == 2:
        c = torch.randn(M, N, generator=g, device=dev)
        c0 = c.clone()
    else:
        c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c), N, _lib.ptr(bias),
                                _lib.ptr(gamma) if epi == 2 else None, _lib.current_stream()), "gemm")
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(16000, 16300, device=dev), torch.arange(M - 700, M, device=dev)])
    ref = a[rows].float() @ b.float().T + bias
    tol = 3e-3 * math.sqrt(K / 64)
    if epi == 0:
        assert (c[rows].float() - ref).abs().max().item() <= tol + 0.01 * ref.abs().max().item()
    elif epi == 1:
        assert (c[rows].float() - torch.nn.functional.gelu(ref)).abs().max().item() <= tol + 0.01 * ref.abs().max().item()
    else:
        assert (c[rows] - (c0[rows] + gamma * ref)).abs().max().item() <= 2 * tol
    # run to run: the same bits
    if epi == 2:
        c2 = c0.clone()
        _lib.check(lib.revo_op_gemm(epi, _lib.ptr(a), K, _lib.ptr(b), K, M, N, K, _lib.ptr(c2), N, _lib.ptr(bias), _lib.ptr(gamma),
                                    _lib.current_stream()), "gemm")
        torch.cuda.synchronize()
        assert torch.equal(c2, c)
