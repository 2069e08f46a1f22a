"""CPU: the oracle against its committed golden vectors and against known-answer
cases that need no oracle at all (SURVEY.md §7 stage 0)."""
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd
from oracle import pe_vit, search as osearch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


def _load(name):
    return np.load(os.path.join(HERE, "golden", name))


def test_tiny_vit_matches_golden():
    for fname, cname in (("tiny_vit.npz", "PE-Tiny-T14-56"), ("tiny_vit_ls.npz", "PE-Tiny-T14-56-LS"),
                         ("tiny_vit_n14.npz", "PE-Tiny-N14-56")):
        gold = _load(fname)
        cfg, sd, images = make_golden.tiny_case(cname)
        assert np.array_equal(images.numpy(), gold["images"])
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        assert abs(wsum - float(gold["weight_abs_sum"])) <= 1e-6 * wsum
        taps = {}
        with torch.no_grad():
            emb = pe_vit.l2_normalize(pe_vit.encode_image(sd, cfg, images, taps))
        for k, v in taps.items():
            np.testing.assert_allclose(v.numpy(), gold["tap_" + k], rtol=0, atol=2e-5)
        np.testing.assert_allclose(emb.numpy(), gold["embedding"], rtol=0, atol=1e-6)
        # fp64 evaluation of the same restatement agrees: the fp32 path is not ill-conditioned
        e64 = pe_vit.embed(sd, cfg, images, dtype=torch.float64)
        assert (e64.float() - emb).abs().max().item() < 1e-5


def test_b16_block_matches_golden():
    gold = _load("b16_block.npz")
    cfg, sd, images = make_golden.b16_block_case()
    assert abs(images.double().sum().item() - float(gold["image_sum"])) < 1e-6
    taps = {}
    with torch.no_grad():
        emb = pe_vit.l2_normalize(pe_vit.encode_image(sd, cfg, images, taps))
    rows = gold["rows"].tolist()
    np.testing.assert_allclose(taps["ln_pre"][0, rows].numpy(), gold["ln_pre"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(taps["block0"][0, rows].numpy(), gold["block0"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(emb.numpy(), gold["embedding"], atol=1e-6, rtol=0)


def test_search_matches_golden():
    gold = _load("search_4096x1024.npz")
    gal, qr, perm = make_golden.search_case()
    assert abs(gal.astype(np.float64).sum() - float(gold["gallery_sum"])) < 1e-6
    for k in (1, 5, 10, 50):
        for thr, tag in ((None, "none"), (0.7, "0p7")):
            s, i, c = osearch.search(gal, qr, k, thr)
            t = f"k{k}_thr{tag}"
            assert np.array_equal(i, gold[t + "_indices"])
            assert np.array_equal(c, gold[t + "_counts"])
            np.testing.assert_allclose(s, gold[t + "_scores"], atol=1e-6, rtol=0)
            # the chunked form (the one the full-size GPU test runs on the 1 M-row gallery) is the same oracle:
            # the committed goldens pin it too
            gn = osearch.normalize_rows(gal)
            cs, ci, cc = osearch.search_chunked([(st, gn[st:st + 1000]) for st in range(4000, -1, -1000)], qr, k, thr)
            assert np.array_equal(ci, gold[t + "_indices"]) and np.array_equal(cc, gold[t + "_counts"])
            assert np.array_equal(cs, s)
    # planted neighbours are found first; duplicate group comes back index-ascending
    i10 = gold["k10_thrnone_indices"]
    assert np.array_equal(i10[0], np.arange(100, 110))
    assert all(i10[q, 0] == perm[q] for q in range(1, 7))
    assert gold["k10_thr0p7_counts"][-1] == 0                 # unrelated query: empty result
    assert np.array_equal(gold["merged_indices"], gold["k10_thrnone_indices"])   # shard+merge == unsharded
    np.testing.assert_allclose(gold["merged_scores"], gold["k10_thrnone_scores"], atol=1e-6)


def test_rope_identity_at_cls_and_norm_preserving():
    cfg = reverso_amd.get_config("PE-Tiny-T14-56")
    ang = pe_vit.rope_angles(cfg, torch.float64)
    assert ang.shape == (cfg.seq, cfg.head_dim)
    assert torch.all(ang[0] == 0)                              # cls sits at (0,0): unrotated
    x = torch.randn(1, 2, cfg.seq, cfg.head_dim, dtype=torch.float64)
    y = pe_vit.apply_rope(x, ang)
    assert torch.equal(y[:, :, 0], x[:, :, 0])
    torch.testing.assert_close(y.norm(dim=-1), x.norm(dim=-1))
    # first patch is at (1,1): x-half and y-half carry the same angles
    hd = cfg.head_dim
    assert torch.equal(ang[1, : hd // 2], ang[1, hd // 2:])
    # interleaved pairs share one frequency
    assert torch.equal(ang[:, 0::2], ang[:, 1::2])


def test_rope_pair_rotation_equals_an_independent_implementation():
    """The one sub-block of the oracle that was pinned only to its own identities: the interleaved-pair rotation.
    ``transformers`` ships GPT-J's ``rotate_every_two`` / ``apply_rotary_pos_emb`` -- the same convention ((x0, x1) ->
    (-x1, x0), one angle per consecutive pair), written independently of this repository.  Exact equality in fp64 on
    random tensors, for every variant's angle table.  (What stays recall is the 2-D LAYOUT of the table -- x half, y
    half, +1 grid offset, zero row for cls -- not the rotation.)"""
    gptj = pytest.importorskip("transformers.models.gptj.modeling_gptj")
    g = torch.Generator().manual_seed(11)
    for name in ("PE-Tiny-T14-56", "PE-Core-B16-224", "PE-Core-L14-336", "PE-Core-G14-448"):
        cfg = reverso_amd.get_config(name)
        ang = pe_vit.rope_angles(cfg, torch.float64)                       # [S, hd], pairs share one angle
        x = torch.randn(2, 3, cfg.seq, cfg.head_dim, generator=g, dtype=torch.float64)      # [B, H, S, hd]
        assert torch.equal(pe_vit.rotate_pairs(x), gptj.rotate_every_two(x))
        half = ang[:, 0::2]
        assert torch.equal(half, ang[:, 1::2])
        sin, cos = half.sin()[None].expand(2, -1, -1), half.cos()[None].expand(2, -1, -1)   # [B, S, hd / 2]
        want = gptj.apply_rotary_pos_emb(x.permute(0, 2, 1, 3), sin, cos).permute(0, 2, 1, 3)   # GPT-J: [B, S, H, hd]
        assert torch.equal(pe_vit.apply_rope(x, ang), want)


def test_known_answer_search_cases():
    D = 64
    eye = np.eye(D, dtype=np.float32)
    s, i, c = osearch.search(eye, eye[:3], 1)
    assert np.array_equal(i[:, 0], [0, 1, 2]) and np.allclose(s[:, 0], 1.0)
    s, i, c = osearch.search(np.concatenate([eye, -eye]), eye[:1], 2 * D)
    assert i[0, 0] == 0 and i[0, -1] == D and np.isclose(s[0, -1], -1.0)      # negation scores -1, last
    # permuting gallery rows permutes the indices
    rng = np.random.default_rng(0)
    g = rng.standard_normal((300, D)).astype(np.float32)
    q = rng.standard_normal((4, D)).astype(np.float32)
    p = rng.permutation(300)
    s0, i0, _ = osearch.search(g, q, 7)
    s1, i1, _ = osearch.search(g[p], q, 7)
    assert np.array_equal(p[i1], i0) and np.allclose(s0, s1, atol=1e-6)
    # empty gallery, k larger than the gallery
    s, i, c = osearch.search(np.zeros((0, D), np.float32), q, 5)
    assert np.all(c == 0) and np.all(i == -1)
    s, i, c = osearch.search(g[:3], q, 5)
    assert np.all(c == 3) and np.all(i[:, 3:] == -1) and np.all(np.isinf(s[:, 3:]))
    # reference-style single query walk agrees with the batched oracle
    gn = osearch.normalize_rows(g)
    ref = osearch.search_one_reference_style(gn, q[0], 7, None)
    assert [r[0] for r in ref] == i0[0].tolist()


def test_layernorm_and_softmax_known_answers():
    x = torch.full((2, 8), 3.0)
    w, b = torch.randn(8), torch.randn(8)
    assert torch.equal(pe_vit.layer_norm(x, w, b, 1e-5), b.expand(2, 8))
    e = torch.randn(5, 16)
    n = pe_vit.l2_normalize(e)
    torch.testing.assert_close(n.norm(dim=-1), torch.ones(5))


# ---- resize oracle pinned to Pillow (the library the reference's preprocess calls) ----------
@pytest.mark.parametrize("h,w,size,box", [(480, 640, 336, None), (100, 50, 56, None), (56, 56, 56, None),
                                          (37, 41, 224, None), (500, 333, 56, (10, 20, 200, 300)),
                                          (64, 64, 56, (5, 7, 6, 8)), (720, 1280, 224, (100, 50, 1100, 700)),
                                          (33, 77, 336, (3, 2, 70, 31)), (57, 3, 56, None)])
def test_resize_oracle_matches_pillow(h, w, size, box):
    from PIL import Image
    from oracle import resize as R
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pil = Image.fromarray(img)
    if box is not None:
        pil = pil.crop(box)
    want = np.asarray(pil.resize((size, size), Image.BILINEAR))
    got = R.crop_resize_u8(img, size, box)
    assert np.array_equal(got, want)


def test_resize_oracle_matches_product_host_resize():
    """preprocess.resize_u8 (host PIL path of the facade) == oracle, CHW vs HWC."""
    from oracle import resize as R
    from reverso_amd import preprocess
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (123, 211, 3), dtype=np.uint8)
    chw = preprocess.resize_u8(img, 56).numpy()
    assert np.array_equal(chw.transpose(1, 2, 0), R.crop_resize_u8(img, 56))


# ---- oracle sub-blocks pinned to the torch modules the upstream tower is assembled from --------
def test_oracle_attention_pool_matches_nn_multiheadattention():
    """The upstream attention-pool head calls torch.nn.MultiheadAttention (probe as query, tokens as
    key/value).  That module is installed here: the oracle's hand-assembled attention must equal it."""
    cfg = reverso_amd.get_config("PE-Tiny-T14-56")
    sd = reverso_amd.weights.synth_weights(cfg, seed=3, randomize_affine=True)
    W, H = cfg.width, cfg.pool_heads
    pre = "visual.attn_pool."
    mha = torch.nn.MultiheadAttention(W, H, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(sd[pre + "attn.in_proj_weight"])
        mha.in_proj_bias.copy_(sd[pre + "attn.in_proj_bias"])
        mha.out_proj.weight.copy_(sd[pre + "attn.out_proj.weight"])
        mha.out_proj.bias.copy_(sd[pre + "attn.out_proj.bias"])
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, cfg.seq, W, generator=g)
    probe = sd[pre + "probe"].reshape(1, 1, W).expand(3, 1, W)
    with torch.no_grad():
        o, _ = mha(probe, x, x, need_weights=False)
        ln = torch.nn.functional.layer_norm(o, (W,), sd[pre + "layernorm.weight"], sd[pre + "layernorm.bias"], cfg.ln_eps)
        h = torch.nn.functional.gelu(torch.nn.functional.linear(ln, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"]))
        want = (o + torch.nn.functional.linear(h, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"]))[:, 0]
    got = pe_vit.attn_pool(x, sd, cfg)
    assert (got - want).abs().max().item() <= 2e-5


def test_oracle_self_attention_matches_sdpa_and_mha_without_rope():
    """With the rotation switched off (angle 0) the body attention is plain multi-head attention:
    equal to torch.nn.MultiheadAttention and to F.scaled_dot_product_attention on the same weights."""
    cfg = reverso_amd.get_config("PE-Tiny-T14-56")
    sd = reverso_amd.weights.synth_weights(cfg, seed=4, randomize_affine=True)
    W, H = cfg.width, cfg.heads
    pre = "visual.transformer.resblocks.0."
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, cfg.seq, W, generator=g)
    zero = torch.zeros(cfg.seq, W // H)
    got = pe_vit.self_attention(x, sd, cfg, pre, zero)
    mha = torch.nn.MultiheadAttention(W, H, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(sd[pre + "attn.in_proj_weight"])
        mha.in_proj_bias.copy_(sd[pre + "attn.in_proj_bias"])
        mha.out_proj.weight.copy_(sd[pre + "attn.out_proj.weight"])
        mha.out_proj.bias.copy_(sd[pre + "attn.out_proj.bias"])
        want, _ = mha(x, x, x, need_weights=False)
    assert (got - want).abs().max().item() <= 2e-5
    # and the rotation itself is norm preserving and relative: <rope(q, i), rope(k, j)> depends on i - j only
    ang = pe_vit.rope_angles(cfg)
    q = torch.randn(1, 1, cfg.seq, W // H, generator=g)
    rq = pe_vit.apply_rope(q, ang)
    assert torch.allclose(rq.norm(dim=-1), q.norm(dim=-1), atol=1e-5)
