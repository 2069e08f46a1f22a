"""CPU oracle: fp32 restatement of the Perception-Encoder image tower.

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg; never by the product package.

PARITY UNPINNED: the arithmetic lives in facebookresearch/perception_models,
cloned at un-pinned HEAD by the reference (``setup.sh:230``), absent from
``/root/reference`` and from this image, and the reference holds no tests or
golden vectors for it.  This file restates the published
architecture as recalled in SURVEY.md §8(a); it follows the reference's own
call sites:

* ``core_system.py:439``   preprocess(...).unsqueeze(0)  -> ``images`` is [B,3,H,W] fp32 in [-1,1]
* ``core_system.py:441-442`` ``pe_model.encode_image(x)`` under ``no_grad``  -> ``encode_image``
* ``core_system.py:443-445`` 2-D features are used as is (3-D would be token-mean)
* ``core_system.py:447``   ``embedding / embedding.norm()`` (no epsilon)       -> ``l2_normalize``

Weight names are the upstream checkpoint names (``visual.*``), so a real
checkpoint can be dropped in and this file re-validated against upstream.
Everything is plain ``torch`` ops on CPU tensors; ``dtype`` may be float32 (the
reference's defining path) or float64 (tight checker).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- 2-D RoPE ---
def rope_angles(cfg, dtype=torch.float32):
    """Angle table [S, head_dim] for the axial 2-D rotary embedding.

    Upstream ``Rope2D``: per axis a rotary embedding of dim head_dim/2 with
    freqs ``theta ** -(arange(0, d, 2)/d)``, each frequency repeated for an
    interleaved pair; x-axis angles fill the first half of head_dim, y-axis the
    second; patch coordinates start at 1 when a cls token exists and the cls
    row is all zeros, i.e. cls is left unrotated.
    """
    hd = cfg.width // cfg.heads
    d = hd // 2
    g = cfg.image_size // cfg.patch_size
    freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    start = 1 if cfg.use_cls else 0
    pos = torch.arange(g, dtype=torch.float64) + start
    ang = pos[:, None] * freqs[None, :]                 # [g, d/2]
    ang = ang.repeat_interleave(2, dim=-1)              # [g, d]  (f0,f0,f1,f1,...)
    ay = ang[:, None, :].expand(g, g, d)                # row (y) position
    ax = ang[None, :, :].expand(g, g, d)                # column (x) position
    tab = torch.cat([ax, ay], dim=-1).reshape(g * g, hd)
    if cfg.use_cls:
        tab = torch.cat([torch.zeros(1, hd, dtype=torch.float64), tab], dim=0)
    return tab.to(dtype)


def rotate_pairs(x):
    """(x0,x1,x2,x3,..) -> (-x1,x0,-x3,x2,..): interleaved-pair rotation."""
    x = x.unflatten(-1, (-1, 2))
    x0, x1 = x.unbind(-1)
    return torch.stack((-x1, x0), dim=-1).flatten(-2)


def apply_rope(x, ang):
    """x: [B,H,S,hd], ang: [S,hd]."""
    return x * ang.cos() + rotate_pairs(x) * ang.sin()


# ------------------------------------------------------------- sub-blocks ---
def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def self_attention(x, p, cfg, prefix, ang):
    """Fused in-proj -> heads -> RoPE(q,k) -> softmax(q k^T / sqrt(hd)) v -> out-proj."""
    B, S, W = x.shape
    H = cfg.heads
    hd = W // H
    qkv = F.linear(x, p[prefix + "attn.in_proj_weight"], p[prefix + "attn.in_proj_bias"])
    q, k, v = qkv.unflatten(-1, (3, W)).unbind(-2)
    q = q.reshape(B, S, H, hd).transpose(1, 2)
    k = k.reshape(B, S, H, hd).transpose(1, 2)
    v = v.reshape(B, S, H, hd).transpose(1, 2)
    q = apply_rope(q, ang)
    k = apply_rope(k, ang)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, S, W)
    return F.linear(o, p[prefix + "attn.out_proj.weight"], p[prefix + "attn.out_proj.bias"])


def mlp(x, p, prefix):
    h = F.linear(x, p[prefix + "c_fc.weight"], p[prefix + "c_fc.bias"])
    h = F.gelu(h)                                   # exact (erf) GELU
    return F.linear(h, p[prefix + "c_proj.weight"], p[prefix + "c_proj.bias"])


def attn_pool(x, p, cfg):
    """Attention-pool head: one learned probe, nn.MultiheadAttention semantics
    (separate q/k/v slices of in_proj, scale hd^-1/2, out_proj), then
    ``x + mlp(layernorm(x))``."""
    B, S, W = x.shape
    H = cfg.pool_heads
    hd = W // H
    pre = "visual.attn_pool."
    wi, bi = p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"]
    probe = p[pre + "probe"].reshape(1, 1, W).expand(B, 1, W)
    q = F.linear(probe, wi[:W], bi[:W]).reshape(B, 1, H, hd).transpose(1, 2)
    k = F.linear(x, wi[W:2 * W], bi[W:2 * W]).reshape(B, S, H, hd).transpose(1, 2)
    v = F.linear(x, wi[2 * W:], bi[2 * W:]).reshape(B, S, H, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, 1, W)
    o = F.linear(o, p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"])
    h = layer_norm(o, p[pre + "layernorm.weight"], p[pre + "layernorm.bias"], cfg.ln_eps)
    o = o + mlp(h, p, pre + "mlp.")
    return o[:, 0]


# --------------------------------------------------------------- forward ---
def encode_image(params, cfg, images, taps=None):
    """``pe_model.encode_image`` (called at core_system.py:341, :442).

    images: [B,3,H,W] already preprocessed to [-1,1].  Returns the
    UN-normalised [B, out_dim] features (upstream ``normalize=False`` default,
    which is why the reference normalises itself at core_system.py:447).
    ``taps``: optional dict that receives named intermediate activations.
    """
    p = params
    dt = images.dtype
    B = images.shape[0]
    P, W = cfg.patch_size, cfg.width

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().clone()

    x = F.conv2d(images, p["visual.conv1.weight"], None, stride=P)       # [B,W,g,g]
    x = x.flatten(2).transpose(1, 2)                                      # [B,g*g,W]
    if cfg.use_cls:
        cls = p["visual.class_embedding"].reshape(1, 1, W).expand(B, 1, W)
        x = torch.cat([cls, x], dim=1)
    x = x + p["visual.positional_embedding"][None]
    tap("embed", x)
    x = layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"], cfg.ln_eps)
    tap("ln_pre", x)
    ang = rope_angles(cfg, dt).to(images.device)      # (the oracle also runs on a device tensor: tests/test_gpu_l14_error_budget.py)
    for i in range(cfg.layers):
        pre = f"visual.transformer.resblocks.{i}."
        h = layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], cfg.ln_eps)
        a = self_attention(h, p, cfg, pre, ang)
        if cfg.use_ls:
            a = a * p[pre + "ls_1.gamma"]
        x = x + a
        h = layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], cfg.ln_eps)
        m = mlp(h, p, pre + "mlp.")
        if cfg.use_ls:
            m = m * p[pre + "ls_2.gamma"]
        x = x + m
        tap(f"block{i}", x)
    x = layer_norm(x, p["visual.ln_post.weight"], p["visual.ln_post.bias"], cfg.ln_eps)
    tap("ln_post", x)
    pooled = attn_pool(x, p, cfg)
    tap("pooled", pooled)
    out = pooled @ p["visual.proj"]
    tap("proj", out)
    return out


def l2_normalize(e):
    """core_system.py:447 ``embedding / embedding.norm()`` — no epsilon."""
    return e / e.norm(dim=-1, keepdim=True)


@torch.no_grad()
def embed(params, cfg, images, dtype=torch.float32):
    """images [B,3,H,W] in [-1,1] -> L2-normalised [B, out_dim] (process_image_direct_pe,
    core_system.py:439-448), batched."""
    p = {k: v.to(dtype) for k, v in params.items()}
    return l2_normalize(encode_image(p, cfg, images.to(dtype)))


@torch.no_grad()
def embed_batch1(params, cfg, images):
    """The reference's actual execution: one image per forward (``unsqueeze(0)``,
    core_system.py:439) in fp32.  Used for the cpu_baseline timing."""
    outs = [l2_normalize(encode_image(params, cfg, images[i:i + 1])) for i in range(images.shape[0])]
    return torch.cat(outs, 0)


def preprocess_u8(images_u8):
    """uint8 [B,3,H,W] already at model resolution -> fp32 in [-1,1]:
    ToTensor (x/255) then Normalize(mean 0.5, std 0.5) (transform built at
    core_system.py:200)."""
    return (images_u8.to(torch.float32) / 255.0 - 0.5) / 0.5
