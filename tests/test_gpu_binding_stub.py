"""The reference-side binding of INTEGRATION.md section 2, executed as it stands, and the fp32-image input mode at full size.

What a maintainer of the reference would paste over ``core_system.py:341/:442`` (``pe_model.encode_image``), ``:447``
(``embedding / embedding.norm()``), ``:600-622`` (collection create + upsert) and ``:659-664`` (``vector_db.search``) is
the fenced block of INTEGRATION.md section 2.  This file extracts that block from the document and runs it verbatim
(only ``REVO_LIB`` in the environment says where the library is): ``make_vit`` -> ``encode_image_normalized`` on **fp32**
``[B,3,336,336]`` tensors -- what ``self.preprocess(...)`` yields at ``core_system.py:335/:439`` -- -> ``make_gallery`` ->
``upsert`` -> ``search``, against the committed oracle goldens of PE-Core-L14-336 (all 64 images of the headline batch).

The fp32-image mode takes the three-part split-precision patch GEMM (hi | hi | lo of 255 x against
hi | lo | hi of w / 255), which the uint8 tests never reach: here at L14 (K = 588 -> 640), B16 (patch 16, K = 768) and G14.

Tolerances as everywhere (tests/_parity.py): cosine >= 0.9999, centred cosine >= 0.99, cosine scores against a 2000-row probe
gallery within 1e-3 on EVERY pair, top-k indices equal to the oracle's on the same stored vectors."""
import os
import re
import sys

import numpy as np
import pytest
import torch

import reverso_amd
from reverso_amd import _lib, engine, weights
from oracle import pe_vit
from oracle import search as osearch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import make_golden_l14 as mg  # noqa: E402
import make_golden_g14 as mg14  # noqa: E402
from _parity import assert_embeddings_match  # noqa: E402


def _stub_source():
    """The first fenced python block under '## 2.' of INTEGRATION.md, character for character."""
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    sec = text[text.index("## 2. Binding stub"):]
    sec = sec[: sec.index("\n## 3.")]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 2 has no fenced python block"
    return m.group(1)


def _probe_gallery(D, n=2000, seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=-1)


def test_integration_stub_runs_verbatim_against_the_l14_goldens(dev, monkeypatch):
    lib_path = os.path.join(ROOT, "revers-o_amd", "librevo.so")
    assert os.path.exists(lib_path), "librevo.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    monkeypatch.setenv("REVO_LIB", lib_path)
    src = _stub_source()
    assert "revo_vit_forward" in src and "revo_search_topk" in src and "revo_gallery_append" in src
    ns = {}
    exec(compile(src, "INTEGRATION.md#2", "exec"), ns)           # the maintainer's paste, as documented

    gold = np.load(os.path.join(HERE, "golden", "l14_batch64.npz"))
    cfg, sd, u8 = mg.batch_case()
    assert int(u8.long().sum()) == int(gold["image_sum"])
    # the reference's own input type: ToTensor + Normalize(0.5, 0.5) on the host, fp32 CHW in [-1, 1] (core_system.py:335/:439)
    x = pe_vit.preprocess_u8(u8)
    assert x.dtype == torch.float32 and tuple(x.shape) == (64, 3, 336, 336)
    state = {k: v.to(dev) for k, v in sd.items()}
    state["text.unused"] = torch.zeros(3, device=dev)             # a whole-CLIP state dict: the stub keeps `visual.*` only
    vit = ns["make_vit"](state, max_batch=64)
    emb = ns["encode_image_normalized"](vit, x.to(dev))
    torch.cuda.synchronize()
    emb = emb.cpu()
    ref = torch.from_numpy(gold["embedding"])
    assert gold["idx"].tolist() == list(range(64))
    stats = assert_embeddings_match(emb, ref, what="INTEGRATION.md stub, fp32 images")
    assert stats["cosine_min"] >= 0.99999 and stats["centred_cosine_min"] >= 0.999, stats
    gal = _probe_gallery(cfg.out_dim)
    assert ((emb @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3                      # all 64 x 2000 pairs
    assert ((emb.norm(dim=-1) - 1).abs() <= 1e-5).all()

    # u8 and fp32 forms of the same pixels: the embedded tokens agree to ~1e-6, the embeddings to bf16 noise
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=64)
    e_u8 = eng.embed(u8.to(dev)).cpu()
    e_f32 = eng.embed(x.to(dev)).cpu()
    assert torch.equal(e_f32, emb)                                # the stub and VitEngine are the same calls
    assert ((e_u8 * e_f32).sum(-1) >= 0.99999).all() and (e_u8 - e_f32).abs().max().item() <= 1e-3
    eng.close()

    # gallery: the probe rows + the 64 oracle embeddings, upserted in two pieces (host and device sources)
    rows = torch.cat([gal, ref])
    gh = ns["make_gallery"](cfg.out_dim, rows.shape[0] + 8)
    ns["upsert"](gh, rows[:1000])                                 # host tensor
    ns["upsert"](gh, rows[1000:].to(dev))                         # device tensor
    stored = osearch.normalize_rows(rows.numpy())
    for i in (0, 17, 63):
        for k, thr in ((5, 0.0), (10, 0.5), (3, 0.99999)):
            got = ns["search"](gh, emb[i], k, thr)
            rs, ri, rc = osearch.search(stored, emb[i].numpy()[None], k, thr, normalize=True)
            n = int(rc[0])
            assert [j for j, _ in got] == ri[0, :n].tolist(), (i, k, thr)
            assert np.abs(np.array([s for _, s in got]) - rs[0, :n]).max(initial=0.0) <= 1e-5
        best = ns["search"](gh, emb[i], 1, 0.0)
        assert best[0][0] == 2000 + i and best[0][1] >= 0.9999    # the image's own oracle embedding
    lib = _lib.load()
    lib.revo_gallery_destroy(gh)
    lib.revo_vit_destroy(vit)


def test_b16_patch16_fp32_images_full_depth_vs_oracle(dev):
    """PE-Core-B16-224 (patch 16: K = 3 x 256 = 768 per part, no padding columns) with fp32 images, all 12 blocks."""
    cfg = reverso_amd.get_config("PE-Core-B16-224")
    sd = weights.synth_weights(cfg, seed=0, randomize_affine=True)
    g = torch.Generator().manual_seed(1234)
    u8 = torch.randint(0, 256, (6, 3, 224, 224), generator=g, dtype=torch.uint8)
    x = pe_vit.preprocess_u8(u8)
    ref = pe_vit.embed(sd, cfg, x)
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=8)
    e_f32 = eng.embed(x.to(dev)).cpu()
    assert_embeddings_match(e_f32, ref, what="B16 fp32 images")
    gal = _probe_gallery(cfg.out_dim)
    assert ((e_f32 @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    e_u8 = eng.embed(u8.to(dev)).cpu()
    assert ((e_u8 * e_f32).sum(-1) >= 0.99999).all() and (e_u8 - e_f32).abs().max().item() <= 1e-3
    # float images that are NOT on the uint8 lattice (a resized photograph after ToTensor is, a tensor a caller made is not)
    xr = (torch.rand(6, 3, 224, 224, generator=g) * 2 - 1)
    refr = pe_vit.embed(sd, cfg, xr)
    assert_embeddings_match(eng.embed(xr.to(dev)).cpu(), refr, what="B16 fp32 images off the lattice")
    eng.close()
    # the embedded tokens themselves (before ln_pre): the split-precision patch GEMM against the oracle's conv, fp32 input
    engx = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=8, experiments=True)
    tok = engx.residual_after(xr.to(dev), -2).cpu()
    taps = {}
    with torch.no_grad():
        pe_vit.encode_image(sd, cfg, xr, taps)
    r = taps["embed"]
    assert float((tok - r).norm() / r.norm()) <= 2e-5, float((tok - r).norm() / r.norm())
    engx.close()


def test_g14_fp32_images_vs_golden(dev):
    """configs[4]'s tower with fp32 images: the two golden images of tests/golden/g14_batch32.npz."""
    gold = np.load(os.path.join(HERE, "golden", "g14_batch32.npz"))
    cfg, sd, u8 = mg14.batch_case()
    assert int(u8.long().sum()) == int(gold["image_sum"])
    gi = gold["images"].tolist()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2)
    del sd
    x = pe_vit.preprocess_u8(u8[gi])
    e = eng.embed(x.to(dev)).cpu()
    ref = torch.from_numpy(gold["embedding"])
    assert_embeddings_match(e, ref, centred_min=-1.0, what="G14 fp32 images")
    gal = _probe_gallery(cfg.out_dim)
    assert ((e @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    e_u8 = eng.embed(u8[gi].to(dev)).cpu()
    assert ((e_u8 * e).sum(-1) >= 0.99999).all()
    eng.close()
