"""Weight sets for the PE vision tower, keyed by the upstream checkpoint names.

The reference loads pretrained weights through ``pe.CLIP.from_config(name,
pretrained=True)`` (``core_system.py:181``), which needs the network.  Here a
state dict (``visual.*`` names) comes either from a file the
user supplies (safetensors / torch) or from the seeded synthetic initialiser the
benchmarks and tests use.
"""
import torch

from .config import PEConfig


def expected_shapes(cfg: PEConfig):
    W, M, D, P, S = cfg.width, cfg.mlp_dim, cfg.out_dim, cfg.patch_size, cfg.seq
    PM = cfg.pool_mlp_dim          # the pool head's own MLP width (4 * W upstream), not the tower's
    sh = {
        "visual.conv1.weight": (W, 3, P, P),
        "visual.positional_embedding": (S, W),
        "visual.ln_pre.weight": (W,), "visual.ln_pre.bias": (W,),
        "visual.ln_post.weight": (W,), "visual.ln_post.bias": (W,),
        "visual.attn_pool.probe": (1, 1, W),
        "visual.attn_pool.attn.in_proj_weight": (3 * W, W),
        "visual.attn_pool.attn.in_proj_bias": (3 * W,),
        "visual.attn_pool.attn.out_proj.weight": (W, W),
        "visual.attn_pool.attn.out_proj.bias": (W,),
        "visual.attn_pool.layernorm.weight": (W,), "visual.attn_pool.layernorm.bias": (W,),
        "visual.attn_pool.mlp.c_fc.weight": (PM, W), "visual.attn_pool.mlp.c_fc.bias": (PM,),
        "visual.attn_pool.mlp.c_proj.weight": (W, PM), "visual.attn_pool.mlp.c_proj.bias": (W,),
        "visual.proj": (W, D),
    }
    if cfg.use_cls:
        sh["visual.class_embedding"] = (W,)
    for i in range(cfg.layers):
        p = f"visual.transformer.resblocks.{i}."
        sh.update({
            p + "ln_1.weight": (W,), p + "ln_1.bias": (W,),
            p + "ln_2.weight": (W,), p + "ln_2.bias": (W,),
            p + "attn.in_proj_weight": (3 * W, W), p + "attn.in_proj_bias": (3 * W,),
            p + "attn.out_proj.weight": (W, W), p + "attn.out_proj.bias": (W,),
            p + "mlp.c_fc.weight": (M, W), p + "mlp.c_fc.bias": (M,),
            p + "mlp.c_proj.weight": (W, M), p + "mlp.c_proj.bias": (W,),
        })
        if cfg.use_ls:
            sh[p + "ls_1.gamma"] = (W,)
            sh[p + "ls_2.gamma"] = (W,)
    return sh


def synth_weights(cfg: PEConfig, seed: int = 0, device="cpu", randomize_affine: bool = False):
    """Seeded random-init weights of the variant's architecture (fp32).

    ``N(0, 0.02^2)`` for linear/conv weights, LayerNorm gamma=1 beta=0, biases 0,
    probe/cls/pos-emb/proj ``N(0, 1/W)``.  With
    ``randomize_affine`` the LayerNorm affine terms, biases and LayerScale are
    perturbed too so tests exercise every epilogue term.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    W = cfg.width
    out = {}
    for name, shape in expected_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        is_ln = ("ln_" in name or "layernorm" in name)
        if is_ln and leaf == "weight":
            t = torch.ones(shape)
            if randomize_affine:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif is_ln and leaf == "bias":
            t = torch.zeros(shape)
            if randomize_affine:
                t = 0.1 * torch.randn(shape, generator=g)
        elif leaf in ("bias", "in_proj_bias"):
            t = torch.zeros(shape)
            if randomize_affine:
                t = 0.02 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = torch.full(shape, 0.5)
            if randomize_affine:
                t = t + 0.1 * torch.randn(shape, generator=g)
        elif leaf in ("probe", "class_embedding", "positional_embedding", "proj"):
            t = torch.randn(shape, generator=g) * (W ** -0.5)
        else:
            t = torch.randn(shape, generator=g) * 0.02
        out[name] = t.to(device)
    return out


def load_state_dict(path: str):
    """Load a user-supplied checkpoint (safetensors or torch) and keep the
    ``visual.*`` tensors as fp32 CPU tensors."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if "state_dict" in sd:
            sd = sd["state_dict"]
    return {k: v.float() for k, v in sd.items() if k.startswith("visual.")}


def check_state_dict(cfg: PEConfig, sd):
    exp = expected_shapes(cfg)
    missing = [k for k in exp if k not in sd]
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    for k, shp in exp.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
