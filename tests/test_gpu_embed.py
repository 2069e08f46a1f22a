"""Embed parity: the HIP forward (through the C ABI) against the CPU oracle and
the committed golden vectors, layer by layer and end to end.

Tolerances: the product computes matmuls in bf16 with fp32 accumulation and keeps
the residual stream in fp32; the oracle is fp32 throughout.  Stated bounds:
residual stream max|d| <= 3e-2 * max|x| per block, final embedding cosine >= 0.9999 and centred cosine >= 0.99
(tests/_parity.py)
and cosine scores against a gallery within 1e-3 (north_star)."""
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd
from reverso_amd import engine, weights
from oracle import pe_vit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402
sys.path.insert(0, HERE)
from _parity import assert_embeddings_match  # noqa: E402


def _gold(name):
    return np.load(os.path.join(HERE, "golden", name))


@pytest.mark.parametrize("fname,cname", [("tiny_vit.npz", "PE-Tiny-T14-56"), ("tiny_vit_ls.npz", "PE-Tiny-T14-56-LS"),
                                         ("tiny_vit_n14.npz", "PE-Tiny-N14-56")])
def test_tiny_vit_layer_by_layer(dev, fname, cname):
    gold = _gold(fname)
    cfg, sd, images = make_golden.tiny_case(cname)
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=4)
    # intermediate activations come from librevo_exp.so (the product library has no such hooks): same sources, same bits
    engx = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=4, experiments=True)
    img = images.to(dev)
    assert torch.equal(engx.embed(img), eng.embed(img))
    for n, tap in ((0, "ln_pre"), (1, "block0"), (2, "block1")):
        x = engx.residual_after(img, n).cpu().numpy()
        ref = gold["tap_" + tap]
        err = np.abs(x - ref).max()
        assert err <= 3e-2 * np.abs(ref).max(), (tap, err, np.abs(ref).max())
    # the kernels outside the transformer blocks, each against its own golden tap: patchify + patch GEMM + position
    # add + class-token rows (embed), the last LayerNorm (ln_post), the single-probe pool attention + its MLP (pooled)
    taps = engx.taps(img)
    for name in ("embed", "ln_post", "pooled"):
        got, ref = taps[name].cpu().numpy(), gold["tap_" + name]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= 3e-2 * np.abs(ref).max(), (name, err, np.abs(ref).max())
    if cfg.use_cls:      # the class-token row is exactly cls + pos[0] (fp32 adds, no matmul)
        ref0 = (sd["visual.class_embedding"] + sd["visual.positional_embedding"][0]).numpy()
        assert np.abs(taps["embed"][:, 0].cpu().numpy() - ref0[None]).max() <= 1e-6
    emb = eng.embed(img).cpu().numpy()
    ref = gold["embedding"]
    assert_embeddings_match(emb, ref, what=cname)
    assert np.abs(np.linalg.norm(emb, axis=-1) - 1).max() < 1e-5
    un = eng.embed(img, normalize=False).cpu().numpy()
    refun = gold["tap_proj"]
    assert np.abs(un - refun).max() <= 3e-2 * np.abs(refun).max()
    eng.close()
    engx.close()


def test_band_patchify_writes_the_gather_forms_bytes(dev):
    """u8 batches of images whose side is a multiple of 16 are patchified a band of patches per workgroup (pixel rows staged
    in LDS); everything else one output chunk per thread.  Same patch matrix: the embedded tokens (patch GEMM + position, the
    experiment library's tap) of one batch under both forms are equal bit for bit, and so are the embeddings."""
    import os
    for name, side in (("PE-Core-B16-224", 224), ("PE-Core-L14-336", 336)):
        cfg = reverso_amd.get_config(name)
        engx = engine.VitEngine.synthetic(cfg, seed=1, device=0, max_batch=8, experiments=True)
        g = torch.Generator().manual_seed(side)
        u8 = torch.randint(0, 256, (8, 3, side, side), generator=g, dtype=torch.uint8).to(dev)
        tok_band, emb_band = engx.residual_after(u8, -2).clone(), engx.embed(u8).clone()
        os.environ["REVO_PATCHIFY_GATHER"] = "1"
        try:
            tok_gather, emb_gather = engx.residual_after(u8, -2).clone(), engx.embed(u8).clone()
        finally:
            del os.environ["REVO_PATCHIFY_GATHER"]
        assert torch.equal(tok_band, tok_gather) and torch.equal(emb_band, emb_gather), name
        engx.close()


def test_uint8_input_matches_float_preprocess(dev):
    cfg, sd, _ = make_golden.tiny_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=8)
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (5, 3, cfg.image_size, cfg.image_size), generator=g, dtype=torch.uint8)
    e_u8 = eng.embed(u8.to(dev)).cpu()
    e_f = eng.embed(pe_vit.preprocess_u8(u8).to(dev)).cpu()
    # uint8 pixels enter the patch GEMM as exact integers, float images as hi + lo bf16 parts: the embedded tokens agree
    # to ~1e-6, and where that flips a bf16 rounding further down the tower the embeddings differ by bf16 noise
    assert (e_u8 - e_f).abs().max().item() <= 1e-3 and ((e_u8 * e_f).sum(-1) >= 0.99999).all()
    ref = pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(u8))
    assert_embeddings_match(e_u8, ref)
    assert_embeddings_match(e_f, ref)
    # batching: max_batch chunks and batch-1 calls give identical rows
    eng2 = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2)
    e2 = eng2.embed(u8.to(dev)).cpu()
    assert torch.equal(e2, e_u8)
    e1 = torch.cat([eng.embed(u8[i:i + 1].to(dev)).cpu() for i in range(5)])
    assert torch.equal(e1, e_u8)
    eng.close()
    eng2.close()


def test_b16_single_block_golden(dev):
    gold = _gold("b16_block.npz")
    cfg, sd, images = make_golden.b16_block_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=1)
    engx = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=1, experiments=True)
    rows = gold["rows"].tolist()
    x0 = engx.residual_after(images.to(dev), 0).cpu().numpy()[0, rows]
    x1 = engx.residual_after(images.to(dev), 1).cpu().numpy()[0, rows]
    assert np.abs(x0 - gold["ln_pre"]).max() <= 3e-2 * np.abs(gold["ln_pre"]).max()
    assert np.abs(x1 - gold["block0"]).max() <= 3e-2 * np.abs(gold["block0"]).max()
    emb = eng.embed(images.to(dev)).cpu().numpy()
    assert (emb * gold["embedding"]).sum() >= 0.9999
    assert torch.equal(engx.embed(images.to(dev)), eng.embed(images.to(dev)))
    eng.close()
    engx.close()


def test_b16_full_depth_vs_oracle(dev):
    """BASELINE.json configs[0] model (PE-Core-B16-224), 6 images (1182 rows: the 128-row GEMM kernels, the
    separate RoPE kernel and the split-K fc2), all 12 blocks.  The 256 x 256 path with the fused RoPE
    epilogue is covered at the headline size by test_l14_headline_batch_vs_oracle_and_batch_invariance."""
    cfg = reverso_amd.get_config("PE-Core-B16-224")
    sd = weights.synth_weights(cfg, seed=0, randomize_affine=True)
    g = torch.Generator().manual_seed(1234)
    u8 = torch.randint(0, 256, (6, 3, 224, 224), generator=g, dtype=torch.uint8)
    ref = pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(u8))
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=8)
    emb = eng.embed(u8.to(dev)).cpu()
    assert_embeddings_match(emb, ref, what="B16")
    # cosine scores of the embeddings against a fixed random gallery agree to 1e-3
    gal = torch.nn.functional.normalize(torch.randn(2000, cfg.out_dim, generator=g), dim=-1)
    assert ((emb @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    # the same images one at a time take other tile shapes: same answer
    one = torch.cat([eng.embed(u8[i:i + 1].to(dev)).cpu() for i in range(6)])
    assert ((one * emb).sum(-1) >= 0.99995).all()
    eng.close()


def test_g14_shape_family_single_block(dev):
    """PE-Core-G14-448 dimensions (width 1536, 16 heads x 96, MLP 8960, no class token, 1024 tokens, out 1280)
    cut to 2 blocks so the CPU oracle stays cheap: the head_dim-96 attention, K = 1536 / 8960 GEMMs, 192-wide
    pool heads."""
    import dataclasses
    cfg = dataclasses.replace(reverso_amd.get_config("PE-Core-G14-448"), layers=2)
    sd = weights.synth_weights(cfg, seed=5, randomize_affine=True)
    g = torch.Generator().manual_seed(6)
    u8 = torch.randint(0, 256, (2, 3, 448, 448), generator=g, dtype=torch.uint8)
    ref = pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(u8))
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2)
    emb = eng.embed(u8.to(dev)).cpu()
    assert emb.shape == (2, 1280)
    assert_embeddings_match(emb, ref, what="G14 2 blocks")
    eng.close()


def test_embed_rejects_bad_input(dev):
    cfg, sd, _ = make_golden.tiny_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2)
    with pytest.raises(ValueError):
        eng.embed(torch.zeros(1, 3, 64, 64, device=dev))
    with pytest.raises(Exception):
        eng.embed(torch.zeros(1, 3, 56, 56))          # host tensor: no CPU fallback
    bad = dict(sd)
    bad.pop("visual.proj")
    with pytest.raises(KeyError):
        engine.VitEngine(cfg, bad, device=0)
    eng.close()


def test_l14_headline_batch_vs_oracle_and_batch_invariance(dev):
    """The headline configuration itself (PE-Core-L14-336, 64 images per forward: persistent GEMM, split-K
    tail, skinny pool GEMMs, 8-wave attention).  Two of the images are checked against the CPU oracle, and
    every image's embedding must not depend on what else is in the batch (same image alone or in a batch of
    3: other tile shapes, other kernels, same vector up to bf16 noise)."""
    cfg = reverso_amd.get_config("PE-Core-L14-336")
    sd = weights.synth_weights(cfg, seed=0, randomize_affine=True)
    g = torch.Generator().manual_seed(77)
    u8 = torch.randint(0, 256, (64, 3, 336, 336), generator=g, dtype=torch.uint8)
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=64)
    emb = eng.embed(u8.to(dev)).cpu()
    assert torch.isfinite(emb).all()
    assert ((emb.norm(dim=-1) - 1).abs() <= 1e-5).all()
    ref = pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(u8[[0, 63]]))
    assert_embeddings_match(emb[[0, 63]], ref, what="L14 batch 64")
    gal = torch.nn.functional.normalize(torch.randn(2000, cfg.out_dim, generator=g), dim=-1)
    assert ((emb[[0, 63]] @ gal.T) - (ref @ gal.T)).abs().max().item() <= 1e-3
    one = eng.embed(u8[5:6].to(dev)).cpu()
    three = eng.embed(u8[[4, 5, 6]].to(dev)).cpu()
    assert (one[0] * emb[5]).sum() >= 0.99995 and (three[1] * emb[5]).sum() >= 0.99995
    # determinism: the same batch twice gives the same bits (split-K partials are reduced in a fixed order)
    assert torch.equal(eng.embed(u8.to(dev)).cpu(), emb)
    eng.close()


def test_handles_release_device_memory(dev):
    """Create / destroy cycles of the two handle types give their HBM back (no growth over 10 cycles)."""
    from reverso_amd import preprocess
    torch.cuda.synchronize()

    def used():
        free, total = torch.cuda.mem_get_info()
        return total - free

    def cycle():
        eng = engine.VitEngine.synthetic("PE-Tiny-T14-56", seed=0, device=0, max_batch=8)
        img = torch.randint(0, 256, (8, 3, 56, 56), device=dev, dtype=torch.uint8)
        e = eng.embed(img)
        gal = engine.Gallery(eng.cfg.out_dim, 20000, device=0)
        gal.add(torch.randn(20000, eng.cfg.out_dim, device=dev))
        gal.search(e, 5)
        preprocess.crop_resize_device(torch.zeros((100, 120, 3), dtype=torch.uint8, device=dev), None, 56)
        torch.cuda.synchronize()
        gal.close()
        eng.close()

    cycle()                      # first use allocates library-lifetime scratch (preprocess tables, hip modules)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    base = used()
    for _ in range(10):
        cycle()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    assert used() - base <= 8 << 20, (used() - base)


def test_g14_full_depth_sanity(dev):
    """BASELINE.json configs[4]'s tower (PE-Core-G14-448: 50 blocks, width 1536, head_dim 96, no class
    token, S = 1024) end to end: finite unit vectors, and an image's vector does not depend on its batch."""
    cfg = reverso_amd.get_config("PE-Core-G14-448")
    eng = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=8)
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.randint(0, 256, (8, 3, cfg.image_size, cfg.image_size), device=dev, dtype=torch.uint8, generator=g)
    e = eng.embed(img)
    assert torch.isfinite(e).all() and ((e.norm(dim=-1) - 1).abs() <= 1e-5).all()
    one = eng.embed(img[3:4])
    assert float((one[0] * e[3]).sum()) >= 0.99995
    assert torch.equal(eng.embed(img), e)
    eng.close()


@pytest.mark.parametrize("B", [10, 12])
def test_mid_size_batches_are_deterministic(dev, B):
    """Batches whose GEMMs have more 128-row tiles than fit on the chip at once (B16-224, 10-12 images:
    576-684 workgroups): repeated forwards must give the same bits, block by block and at the end."""
    cfg = reverso_amd.get_config("PE-Core-B16-224")
    eng = engine.VitEngine.synthetic(cfg, seed=2, device=0, max_batch=16)
    engx = engine.VitEngine.synthetic(cfg, seed=2, device=0, max_batch=16, experiments=True)
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8).to(dev)
    r = [engx.residual_after(u8, 1).cpu() for _ in range(3)]
    assert torch.equal(r[0], r[1]) and torch.equal(r[0], r[2])
    e = [eng.embed(u8).cpu() for _ in range(3)]
    assert torch.equal(e[0], e[1]) and torch.equal(e[0], e[2]) and torch.equal(e[0], engx.embed(u8).cpu())
    eng.close()
    engx.close()


def test_l14_every_batch_size_family_agrees(dev):
    """PE-Core-L14-336 at batch sizes that land in different GEMM tile families and leftover handlings
    (1, 2, 3, 5, 9, 17, 33 images): every forward is repeatable bit for bit, and an image's vector is the
    same (up to bf16 noise) whatever batch it is embedded in."""
    cfg = reverso_amd.get_config("PE-Core-L14-336")
    eng = engine.VitEngine.synthetic(cfg, seed=0, device=0, max_batch=64, randomize_affine=True)
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (33, 3, 336, 336), generator=g, dtype=torch.uint8).to(dev)
    full = eng.embed(u8)
    for B in (1, 2, 3, 5, 9, 17, 33):
        a = eng.embed(u8[:B])
        b = eng.embed(u8[:B])
        assert torch.equal(a, b), B
        cos = (a * full[:B]).sum(-1)
        assert float(cos.min()) >= 0.99995, (B, float(cos.min()))
    eng.close()


def test_module_level_embed_accepts_pil_images(dev):
    """engine.embed() -- the core API named by north_star -- takes what the reference's process_image_direct_pe takes:
    PIL images (any size: squash-resized like self.preprocess, core_system.py:439), arrays, or a ready device batch."""
    from PIL import Image
    from reverso_amd import preprocess as pp
    cfg, sd, _ = make_golden.tiny_case()
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=4)
    rng = np.random.default_rng(0)
    pils = [Image.fromarray(rng.integers(0, 256, (40 + 9 * i, 70 - 5 * i, 3), dtype=np.uint8)) for i in range(3)]
    a = engine.embed(pils, engine=eng)
    b = eng.embed(pp.batch_u8(pils, cfg.image_size).to(dev))
    assert torch.equal(a, b) and a.shape == (3, cfg.out_dim)
    eng.close()


def test_layerscale_checkpoint_is_auto_detected_and_matches_the_oracle(dev, lib):
    """A checkpoint that carries ls_*.gamma tensors handed to the LayerScale-less variant name: the engine turns LayerScale
    on from the tensors and lands on the oracle run with use_ls = True; an unexpected `visual.*` tensor
    is refused by name, by the Python loader and by the C ABI itself."""
    import ctypes as C
    from reverso_amd import _lib
    gold = _gold("tiny_vit_ls.npz")
    cfg_ls, sd, images = make_golden.tiny_case("PE-Tiny-T14-56-LS")
    plain = reverso_amd.get_config("PE-Tiny-T14-56")
    assert not plain.use_ls and any(".ls_1.gamma" in k for k in sd)
    eng = engine.VitEngine(plain, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=4)
    assert eng.cfg.use_ls
    emb = eng.embed(images.to(dev)).cpu()
    assert_embeddings_match(emb.numpy(), gold["embedding"], what="auto-detected LayerScale")
    with torch.no_grad():
        ref = pe_vit.embed(sd, cfg_ls, images)
        ref_off = pe_vit.embed({k: v for k, v in sd.items() if ".ls_" not in k}, plain, images)
    assert_embeddings_match(emb, ref)
    # ignoring the gains is a different model: the oracle without LayerScale is measurably elsewhere
    assert float(torch.nn.functional.cosine_similarity(emb, ref_off, dim=-1).min()) < 0.9999
    eng.close()
    bad = {k: v.to(dev) for k, v in sd.items()}
    bad["visual.foo"] = torch.zeros(3, device=dev)
    with pytest.raises(KeyError, match=r"visual\.foo"):
        engine.VitEngine(plain, bad, device=0)
    # the C ABI refuses it too: LayerScale tensors with cfg.use_ls = 0, and any tensor the architecture does not read
    for extra, use_ls, msg in (("visual.transformer.resblocks.0.ls_1.gamma", 0, b"unexpected weight tensor"), ("visual.foo", 1, b"visual.foo")):
        names = sorted(k for k in sd if use_ls or ".ls_" not in k) + ([extra] if extra not in sd or not use_ls else [])
        keep = [(sd[n] if n in sd else torch.zeros(plain.width)).float().contiguous() for n in names]
        arr = (_lib.Tensor * len(names))()
        for i, (n, t) in enumerate(zip(names, keep)):
            arr[i].name, arr[i].data, arr[i].numel = n.encode(), t.data_ptr(), t.numel()
        c = _lib.VitCfg(plain.image_size, plain.patch_size, plain.width, plain.layers, plain.heads, plain.mlp_dim, plain.out_dim,
                        plain.pool_heads, 1, use_ls, plain.ln_eps, plain.rope_theta, plain.pool_mlp_dim)
        h = C.c_void_p()
        assert lib.revo_vit_create(C.byref(c), arr, len(names), 0, 2, C.byref(h)) == -2
        assert msg in lib.revo_last_error(), lib.revo_last_error()
