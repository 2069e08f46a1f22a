#!/usr/bin/env python3
"""Pin the oracle to the real reference arithmetic -- the one route from "parity unpinned" to green.

The reference's embed arithmetic lives in facebookresearch/perception_models (cloned at HEAD by setup.sh:230,
imported at core_system.py:28-29) and in a pretrained PE-Core checkpoint.  Neither exists in the build container
(no network; do NOT try to fetch or vendor them).  On a machine that has both, run

    PYTHONPATH=/path/to/perception_models python scripts/make_upstream_fixtures.py \
        --variant PE-Core-L14-336 [--checkpoint /path/to/PE-Core-L14-336.pt] [--out tests/golden]

It runs the UPSTREAM model exactly the way the reference does (core_system.py:181 from_config(pretrained=True),
:195 fp32 on CPU here, :200 get_image_transform, :439-447 preprocess -> encode_image -> e / e.norm()) on two seeded
images and writes  tests/golden/upstream_<variant>.npz  with

    images_u8        [2, H, W, 3]  the source pictures (uint8, larger than the model resolution: the resize is exercised)
    preprocessed     [2, 3, S, S]  what upstream's transform produced (pins oracle/resize.py + the normalisation)
    embedding_raw    [2, D]        encode_image output (un-normalised, upstream default)
    embedding        [2, D]        after the reference's e / e.norm()
    tap_block<i>     [2, S_tok, W] residual stream after transformer block i (forward hooks; every block)
    tap_ln_pre / tap_ln_post / tap_pooled  where the module layout exposes them
    weight_checksums               |w|.sum() of a few named tensors: the tests refuse a different checkpoint
    upstream_commit                `git rev-parse HEAD` of the perception_models checkout, if it is one

and the visual.* state dict as  upstream_<variant>.safetensors  next to it (NOT for committing: ~0.6 GB for L14;
point REVERSO_PE_CHECKPOINT at it).  tests/test_upstream_fixtures.py then checks oracle/pe_vit.py (CPU tier) and
the HIP engine (-m gpu) against these vectors; both skip while the files are absent.
"""
import argparse
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="PE-Core-L14-336")
    ap.add_argument("--checkpoint", default=None, help="local checkpoint; default: upstream's pretrained=True download")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()

    try:
        import core.vision_encoder.pe as pe                         # core_system.py:28
        import core.vision_encoder.transforms as transforms          # core_system.py:29
    except ImportError as e:
        raise SystemExit(f"perception_models is not importable ({e}): put its checkout on PYTHONPATH") from e
    from PIL import Image

    if args.checkpoint:
        model = pe.CLIP.from_config(args.variant, pretrained=False)
        sd = torch.load(args.checkpoint, map_location="cpu")
        sd = sd.get("state_dict", sd)
        model.load_state_dict(sd, strict=False)
    else:
        model = pe.CLIP.from_config(args.variant, pretrained=True)   # core_system.py:181
    model = model.float().eval()
    size = model.image_size
    preprocess = transforms.get_image_transform(size)                # core_system.py:200

    rng = np.random.default_rng(20240601)
    yy, xx = np.mgrid[0:420, 0:560].astype(np.float32)
    imgs = []
    for _ in range(2):
        a = np.zeros((420, 560, 3), np.float32)
        for _ in range(8):
            cx, cy, r = rng.uniform(0, 560), rng.uniform(0, 420), rng.uniform(20, 200)
            a += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))[..., None] * rng.uniform(0, 255, 3)
        a = a / max(a.max() / 255.0, 1.0) + rng.normal(0, 6, a.shape)
        imgs.append(np.clip(a, 0, 255).astype(np.uint8))
    x = torch.stack([preprocess(Image.fromarray(a).convert("RGB")) for a in imgs])   # core_system.py:439

    taps = {}
    vis = model.visual

    def hook(name):
        def fn(_m, _inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            taps[name] = o.detach().float().cpu().numpy()
        return fn
    handles = []
    blocks = getattr(getattr(vis, "transformer", None), "resblocks", None)
    if blocks is not None:
        for i, blk in enumerate(blocks):
            handles.append(blk.register_forward_hook(hook(f"tap_block{i}")))
    for attr, name in (("ln_pre", "tap_ln_pre"), ("ln_post", "tap_ln_post"), ("attn_pool", "tap_pooled")):
        m = getattr(vis, attr, None)
        if isinstance(m, torch.nn.Module):
            handles.append(m.register_forward_hook(hook(name)))
    with torch.no_grad():
        raw = model.encode_image(x)                                  # core_system.py:442
    for h in handles:
        h.remove()
    if raw.dim() == 3:                                               # core_system.py:443-445
        raw = raw.mean(dim=1)
    emb = raw / raw.norm(dim=-1, keepdim=True)                       # core_system.py:447

    vsd = {k: v.detach().float().contiguous() for k, v in model.state_dict().items() if k.startswith("visual.")}
    names = sorted(vsd)
    picks = [names[0], names[len(names) // 2], names[-1], "visual.proj", "visual.conv1.weight"]
    checks = {n: float(vsd[n].double().abs().sum()) for n in picks if n in vsd}
    commit = ""
    try:
        commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.dirname(pe.__file__)), "rev-parse", "HEAD"],
                                         text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        pass
    os.makedirs(args.out, exist_ok=True)
    base = os.path.join(args.out, f"upstream_{args.variant}")
    np.savez_compressed(base + ".npz", images_u8=np.stack(imgs), preprocessed=x.numpy(), embedding_raw=raw.numpy(),
                        embedding=emb.numpy(), weight_checksum_names=np.array(list(checks)),
                        weight_checksum_values=np.array(list(checks.values()), np.float64),
                        upstream_commit=np.array(commit), variant=np.array(args.variant),
                        tensor_shapes=np.array([f"{k}:{tuple(v.shape)}" for k, v in vsd.items()]), **taps)
    from safetensors.torch import save_file
    save_file(vsd, base + ".safetensors")
    print(f"wrote {base}.npz ({len(taps)} taps) and {base}.safetensors ({len(vsd)} tensors; do not commit the latter)")
    print("shapes that decide PEConfig:", {k: tuple(v.shape) for k, v in vsd.items()
                                          if k in ("visual.conv1.weight", "visual.positional_embedding", "visual.proj",
                                                   "visual.attn_pool.mlp.c_fc.weight")})


if __name__ == "__main__":
    sys.exit(main())
