"""CPU: the checkpoint loader (reference: core_system.py:181 `pe.CLIP.from_config(name, pretrained=True)`, whose weights
this build takes from a file).  LayerScale follows the checkpoint's `ls_*` tensors; every other
`visual.*` tensor the architecture table does not name is rejected BY NAME -- a tensor the forward would skip is a
different model; missing tensors and wrong shapes are rejected as before.  Shapes only: meta tensors, no arithmetic."""
import re

import pytest
import torch

import reverso_amd
from reverso_amd import weights


def _meta_sd(cfg):
    return {k: torch.empty(shp, device="meta") for k, shp in weights.expected_shapes(cfg).items()}


def test_layerscale_is_detected_from_the_checkpoint():
    l14 = reverso_amd.get_config("PE-Core-L14-336")
    assert not l14.use_ls
    sd = _meta_sd(l14)
    assert weights.resolve_config(l14, sd) is l14                          # no ls_* tensors: LayerScale off
    weights.check_state_dict(l14, sd)
    for i in range(l14.layers):
        for j in (1, 2):
            sd[f"visual.transformer.resblocks.{i}.ls_{j}.gamma"] = torch.empty((l14.width,), device="meta")
    cfg = weights.resolve_config(l14, sd)
    assert cfg.use_ls and cfg.name == l14.name and cfg.layers == 24
    weights.check_state_dict(cfg, sd)                                      # and the LS-bearing dict is complete for it
    with pytest.raises(KeyError, match=r"ls_1\.gamma"):                    # against the LS-less table they are unexpected
        weights.check_state_dict(l14, sd)
    # a config that says LayerScale with a checkpoint that has none: the checkpoint decides -- and says so loudly (a
    # truncated or wrong checkpoint of an LS variant would otherwise run as a different model without a word)
    tiny_ls = reverso_amd.get_config("PE-Tiny-T14-56-LS")
    with pytest.warns(RuntimeWarning, match="declares LayerScale"):
        assert not weights.resolve_config(tiny_ls, _meta_sd(reverso_amd.get_config("PE-Tiny-T14-56"))).use_ls
    # tensors outside the image tower (a whole-CLIP state dict) are not this table's business
    sd2 = _meta_sd(l14)
    sd2["text.token_embedding.weight"] = torch.empty((8, 8), device="meta")
    sd2["logit_scale"] = torch.empty((), device="meta")
    weights.check_state_dict(l14, sd2)


def test_partial_layerscale_is_rejected():
    l14 = reverso_amd.get_config("PE-Core-L14-336")
    sd = _meta_sd(l14)
    for i in range(l14.layers):
        sd[f"visual.transformer.resblocks.{i}.ls_1.gamma"] = torch.empty((l14.width,), device="meta")
    with pytest.raises(KeyError, match=r"lacks ls_2\.gamma of block 0"):
        weights.resolve_config(l14, sd)
    sd = _meta_sd(l14)
    for i in range(l14.layers + 1):                                        # one block too many: another tower's gains
        for j in (1, 2):
            sd[f"visual.transformer.resblocks.{i}.ls_{j}.gamma"] = torch.empty((l14.width,), device="meta")
    with pytest.raises(KeyError, match="block 24"):
        weights.resolve_config(l14, sd)


def test_unexpected_visual_tensor_is_rejected_by_name():
    l14 = reverso_amd.get_config("PE-Core-L14-336")
    sd = _meta_sd(l14)
    sd["visual.foo"] = torch.empty((3,), device="meta")
    with pytest.raises(KeyError, match=re.escape("visual.foo")):
        weights.check_state_dict(l14, sd)
    del sd["visual.foo"]
    sd["visual.transformer.resblocks.24.ln_1.weight"] = torch.empty((1024,), device="meta")      # a 25th block
    with pytest.raises(KeyError, match=r"resblocks\.24\.ln_1\.weight"):
        weights.check_state_dict(l14, sd)


def test_known_non_parameter_buffers_are_dropped_not_rejected():
    l14 = reverso_amd.get_config("PE-Core-L14-336")
    sd = _meta_sd(l14)
    sd["visual.rope.freqs"] = torch.empty((32,), device="meta")
    sd["visual.transformer.resblocks.3.attn.rope.freqs"] = torch.empty((32,), device="meta")
    weights.check_state_dict(l14, sd)
    kept = weights.strip_non_parameters(sd)
    assert sorted(kept) == sorted(weights.expected_shapes(l14))


def test_missing_and_misshaped_tensors_are_rejected():
    b16 = reverso_amd.get_config("PE-Core-B16-224")
    l14 = reverso_amd.get_config("PE-Core-L14-336")
    sd = _meta_sd(b16)
    with pytest.raises(KeyError, match="lacks"):                           # a B16 checkpoint is not an L14 checkpoint
        weights.check_state_dict(l14, sd)
    sd = _meta_sd(l14)
    sd["visual.proj"] = torch.empty((1024, 768), device="meta")
    with pytest.raises(ValueError, match=r"visual\.proj"):
        weights.check_state_dict(l14, sd)
