"""Golden vectors of the CPU oracle at the headline model size (PE-Core-L14-336), generated in the build
container (minutes of CPU time: the GPU box only reads the .npz files).

    python tests/golden/make_golden_l14.py [batch|outlier|crops ...]

* l14_batch64.npz   oracle embeddings of ALL 64 images of the headline batch (seeded weights and images), and for two
                    of them the residual stream after blocks 6 / 12 / 24, the ln_post output (8 token rows each) and the
                    pooled vector: the error growth through the tower's depth is visible block by block.
* l14_outlier.npz   the same tower with injected outlier channels (a few LayerNorm gains of 20, a few
                    residual-stream channels driven ~100x larger by their out-proj / fc2 rows): the regime real
                    ViT checkpoints are in and N(0, 0.02) synthetic weights are not.
* l14_crops.npz     BASELINE.json configs[2] (detector boxes -> crop -> PE-L14 embed -> search): 64 synthetic
                    JPEGs, 3 boxes each; oracle embedding of every PIL crop().resize(BILINEAR).

Inputs are rebuilt from seeds by the helper functions below (shared with the tests); the files hold the oracle's
outputs plus checksums of the regenerated inputs.  The oracle is "parity unpinned" (oracle/pe_vit.py header).
"""
import io
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import reverso_amd  # noqa: E402
from reverso_amd import weights  # noqa: E402
from oracle import pe_vit  # noqa: E402

VARIANT = "PE-Core-L14-336"
BATCH_IDX = list(range(64))
TAP_IMAGES = [0, 63]                                  # images whose intermediate activations are stored
TAP_TOKENS = [0, 1, 2, 24, 25, 300, 575, 576]         # token rows kept of every tap (class token, corners, middle)
TAP_BLOCKS = [5, 11, 23]


def batch_case():
    """The headline batch: weights seed 0 (affine terms randomised), 64 uint8 images seed 77."""
    cfg = reverso_amd.get_config(VARIANT)
    sd = weights.synth_weights(cfg, seed=0, randomize_affine=True)
    g = torch.Generator().manual_seed(77)
    u8 = torch.randint(0, 256, (64, 3, 336, 336), generator=g, dtype=torch.uint8)
    return cfg, sd, u8


def outlier_case():
    """Weights with the outlier structure of trained ViTs.  In blocks 2, 3, 5 the out-proj and fc2 rows of 3
    output channels are scaled x100 (massive activations in those residual channels from there on), and in every
    4th block a handful of ln_1 / ln_2 gains are 20."""
    cfg = reverso_amd.get_config(VARIANT)
    sd = weights.synth_weights(cfg, seed=1, randomize_affine=True)
    g = torch.Generator().manual_seed(5)
    big = torch.randperm(cfg.width, generator=g)[:3]
    for i in (2, 3, 5):
        p = f"visual.transformer.resblocks.{i}."
        sd[p + "attn.out_proj.weight"][big] *= 100.0
        sd[p + "mlp.c_proj.weight"][big] *= 100.0
    for i in range(0, cfg.layers, 4):
        p = f"visual.transformer.resblocks.{i}."
        ch = torch.randperm(cfg.width, generator=g)[:4]
        sd[p + "ln_1.weight"][ch] = 20.0
        sd[p + "ln_2.weight"][ch] = 20.0
    u8 = torch.randint(0, 256, (4, 3, 336, 336), generator=g, dtype=torch.uint8)
    return cfg, sd, u8, big


N_IMAGES, BOXES_PER_IMAGE, IMG_W, IMG_H = 64, 3, 480, 360


def crops_case():
    """64 synthetic photographs (smooth random fields + texture, JPEG quality 90) and 3 boxes each.
    Returns (list of (filename, jpeg bytes), boxes int32 [64, 3, 4] as inclusive x0, y0, x1, y1)."""
    rng = np.random.default_rng(2024)
    files, boxes = [], np.zeros((N_IMAGES, BOXES_PER_IMAGE, 4), np.int32)
    yy, xx = np.mgrid[0:IMG_H, 0:IMG_W].astype(np.float32)
    for i in range(N_IMAGES):
        img = np.zeros((IMG_H, IMG_W, 3), np.float32)
        for _ in range(6):                                   # a few coloured blobs and gradients
            cx, cy, r = rng.uniform(0, IMG_W), rng.uniform(0, IMG_H), rng.uniform(30, 160)
            col = rng.uniform(0, 255, 3).astype(np.float32)
            w = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
            img += w[..., None] * col
        img = img / max(img.max() / 255.0, 1.0) + rng.normal(0, 8, img.shape).astype(np.float32)
        arr = np.clip(img, 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="JPEG", quality=90)
        files.append((f"img_{i:03d}.jpg", buf.getvalue()))
        for b in range(BOXES_PER_IMAGE):
            w, h = int(rng.integers(40, 300)), int(rng.integers(40, 250))
            x0, y0 = int(rng.integers(0, IMG_W - w)), int(rng.integers(0, IMG_H - h))
            boxes[i, b] = (x0, y0, x0 + w - 1, y0 + h - 1)
    return files, boxes


def crop_weights():
    cfg = reverso_amd.get_config(VARIANT)
    return cfg, weights.synth_weights(cfg, seed=0)          # what SimpleReverso(synthetic_seed=0) builds


def make_batch(path):
    cfg, sd, u8 = batch_case()
    ref = pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(u8[BATCH_IDX]))
    taps = {}
    with torch.no_grad():
        pe_vit.encode_image(sd, cfg, pe_vit.preprocess_u8(u8[TAP_IMAGES]), taps)
    extra = {f"tap_block{b}": taps[f"block{b}"][:, TAP_TOKENS].numpy() for b in TAP_BLOCKS}
    extra["tap_ln_post"] = taps["ln_post"][:, TAP_TOKENS].numpy()
    extra["tap_pooled"] = taps["pooled"].numpy()
    np.savez_compressed(path, idx=np.array(BATCH_IDX), embedding=ref.numpy(), tap_images=np.array(TAP_IMAGES),
                        tap_tokens=np.array(TAP_TOKENS), tap_blocks=np.array(TAP_BLOCKS),
                        image_sum=np.int64(u8.long().sum().item()), **extra)


def make_outlier(path):
    cfg, sd, u8, big = outlier_case()
    taps = {}
    with torch.no_grad():
        out = pe_vit.encode_image(sd, cfg, pe_vit.preprocess_u8(u8), taps)
        emb = pe_vit.l2_normalize(out)
    # how large the outlier channels are in the residual stream (for the record; the test prints it)
    x = taps["block5"]
    np.savez_compressed(path, embedding=emb.numpy(), big_channels=big.numpy(),
                        resid_absmax_big=np.float32(x[..., big].abs().max().item()),
                        resid_absmed=np.float32(x.abs().median().item()), image_sum=np.int64(u8.long().sum().item()))


def make_crops(path):
    cfg, sd = crop_weights()
    files, boxes = crops_case()
    embs, sums = [], []
    for i, (name, data) in enumerate(files):
        im = Image.open(io.BytesIO(data)).convert("RGB")
        sums.append(int(np.asarray(im, dtype=np.int64).sum()))
        crops = []
        for (x0, y0, x1, y1) in boxes[i]:
            c = im.crop((int(x0), int(y0), int(x1) + 1, int(y1) + 1)).resize((cfg.image_size, cfg.image_size), Image.BILINEAR)
            crops.append(torch.from_numpy(np.asarray(c, dtype=np.uint8).transpose(2, 0, 1).copy()))
        embs.append(pe_vit.embed(sd, cfg, pe_vit.preprocess_u8(torch.stack(crops))).numpy())
        print(f"crops {i + 1}/{len(files)}", flush=True)
    np.savez_compressed(path, boxes=boxes, embedding=np.concatenate(embs).astype(np.float32),
                        decoded_sums=np.array(sums, np.int64))


if __name__ == "__main__":
    what = sys.argv[1:] or ["batch", "outlier", "crops"]
    torch.set_num_threads(os.cpu_count() or 1)
    if "batch" in what:
        make_batch(os.path.join(HERE, "l14_batch64.npz"))
    if "outlier" in what:
        make_outlier(os.path.join(HERE, "l14_outlier.npz"))
    if "crops" in what:
        make_crops(os.path.join(HERE, "l14_crops.npz"))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("l14_") and f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
