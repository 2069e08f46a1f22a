"""Where the headline tower's distance from the oracle comes from (PE-Core-L14-336, bf16 operands against the fp32
oracle): the oracle -- the same restatement the goldens were made with, here run by torch in fp32 on the device, first
checked against the committed CPU goldens -- gives every intermediate activation at full size, which the CPU box
could only afford for a handful of vectors.  Per stage: relative L2 distance of the engine's tap from the oracle's;
and the split body / head: the oracle's head (fp32) applied to the ENGINE's ln_post output isolates what the 24 blocks
contribute to the final embedding error, the rest is the attention-pool head and the projection."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import engine
from oracle import pe_vit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_l14 as mg  # noqa: E402


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_l14_error_budget_by_stage(dev):
    gold = np.load(os.path.join(HERE, "golden", "l14_batch64.npz"))
    cfg, sd, u8 = mg.batch_case()
    idx = gold["idx"].tolist()
    sdd = {k: v.to(dev) for k, v in sd.items()}
    imgs = u8[idx].to(dev)
    # the oracle on the device == the oracle on the CPU (the committed goldens)
    taps = {}
    with torch.no_grad():
        ref_out = pe_vit.encode_image(sdd, cfg, pe_vit.preprocess_u8(imgs), taps)
        ref = pe_vit.l2_normalize(ref_out)
    assert (ref.cpu() - torch.from_numpy(gold["embedding"])).abs().max().item() <= 2e-5
    engx = engine.VitEngine(cfg, sdd, device=0, max_batch=len(idx), experiments=True)    # the tap hooks live in librevo_exp.so
    got = engx.taps(imgs)
    budget = {"embed": _rel(got["embed"], taps["embed"])}
    for n in (1, 6, 12, 18, 24):
        budget[f"block{n - 1}"] = _rel(engx.residual_after(imgs, n), taps[f"block{n - 1}"])
    budget["ln_post"] = _rel(got["ln_post"], taps["ln_post"])
    budget["pooled"] = _rel(got["pooled"], taps["pooled"])
    emb = got["embedding"]
    budget["embedding"] = _rel(emb, ref)
    # body vs head: the oracle's fp32 head on the engine's (bf16) ln_post output
    with torch.no_grad():
        hybrid = pe_vit.l2_normalize(pe_vit.attn_pool(got["ln_post"], sdd, cfg) @ sdd["visual.proj"])
        hyb_pool = pe_vit.l2_normalize(got["pooled"] @ sdd["visual.proj"])
    budget["embedding_if_head_were_fp32"] = _rel(hybrid, ref)
    budget["embedding_if_proj_were_fp32"] = _rel(hyb_pool, ref)
    # centred: the common-mode direction of a random-init tower removed
    c_ref, c_emb = ref - ref.mean(0, keepdim=True), emb - emb.mean(0, keepdim=True)
    budget["centred_cosine_min"] = float(torch.nn.functional.cosine_similarity(c_emb, c_ref, dim=-1).min())
    budget["cosine_min"] = float((emb * ref).sum(-1).min())
    budget["pairwise_cosine_between_images"] = float((ref @ ref.T).fill_diagonal_(0).max())
    print("L14 error budget:", json.dumps({k: round(v, 6) for k, v in budget.items()}))
    engx.close()
    assert budget["embedding"] <= 1.2e-2 and budget["cosine_min"] >= 0.9999
