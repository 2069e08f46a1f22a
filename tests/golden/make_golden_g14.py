"""Golden vectors of the CPU oracle for BASELINE.json configs[4]'s tower at full depth: PE-Core-G14-448 (50 blocks,
width 1536, 16 heads x 96, MLP 8960, 1024 tokens, no class token, pool MLP 6144, 1280-dimensional output), generated in
the build container (about a minute of CPU time per image; the GPU box only reads the .npz file).

    python tests/golden/make_golden_g14.py

g14_batch32.npz: the oracle's embedding of images 0 and 31 of the 32-image batch (configs[4]: 256 images over 8 GPUs =
32 per GPU), and for both the residual stream after blocks 10 / 25 / 50 (8 token rows each), the ln_post rows and the
pooled vector.  Inputs are rebuilt from seeds by batch_case() (shared with the test); the file holds a checksum of them.
The oracle is "parity unpinned" (oracle/pe_vit.py header)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import reverso_amd  # noqa: E402
from reverso_amd import weights  # noqa: E402
from oracle import pe_vit  # noqa: E402

VARIANT = "PE-Core-G14-448"
BATCH = 32
GOLD_IMAGES = [0, 31]
TAP_TOKENS = [0, 1, 31, 32, 500, 992, 1022, 1023]
TAP_BLOCKS = [9, 24, 49]


def batch_case(device="cpu"):
    """Weights seed 11 (affine terms randomised), 32 uint8 images seed 12.  The weights can be generated straight on
    the device (3.8 GB in fp32 for this variant)."""
    cfg = reverso_amd.get_config(VARIANT)
    sd = weights.synth_weights(cfg, seed=11, randomize_affine=True)
    g = torch.Generator().manual_seed(12)
    u8 = torch.randint(0, 256, (BATCH, 3, cfg.image_size, cfg.image_size), generator=g, dtype=torch.uint8)
    return cfg, sd, u8


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg, sd, u8 = batch_case()
    taps = {}
    with torch.no_grad():
        out = pe_vit.encode_image(sd, cfg, pe_vit.preprocess_u8(u8[GOLD_IMAGES]), taps)
        emb = pe_vit.l2_normalize(out)
    extra = {f"tap_block{b}": taps[f"block{b}"][:, TAP_TOKENS].numpy() for b in TAP_BLOCKS}
    extra["tap_ln_post"] = taps["ln_post"][:, TAP_TOKENS].numpy()
    extra["tap_pooled"] = taps["pooled"].numpy()
    path = os.path.join(HERE, "g14_batch32.npz")
    np.savez_compressed(path, images=np.array(GOLD_IMAGES), embedding=emb.numpy(), tap_tokens=np.array(TAP_TOKENS),
                        tap_blocks=np.array(TAP_BLOCKS), image_sum=np.int64(u8.long().sum().item()), **extra)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
