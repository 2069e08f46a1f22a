"""Regenerates the committed golden vectors from the CPU oracle.

    python tests/golden/make_golden.py

Inputs are rebuilt from fixed seeds (torch CPU generator / numpy PCG64); the .npz
files hold the oracle's outputs plus checksums of the regenerated inputs so a
generator drift is detected rather than silently compared.  The oracle itself is
"parity unpinned" against the reference (oracle/pe_vit.py header): these files pin
the oracle against regressions and give the GPU box expected values that do not
need the oracle to be re-run at scale.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import reverso_amd  # noqa: E402
from reverso_amd import weights  # noqa: E402
from oracle import pe_vit, search as osearch  # noqa: E402


def tiny_case(name="PE-Tiny-T14-56", seed=11, scale=4.0):
    cfg = reverso_amd.get_config(name)
    sd = weights.synth_weights(cfg, seed=seed, randomize_affine=True)
    # scale the matmul weights up so every block moves the residual stream by O(1)
    for k, v in sd.items():
        if v.dim() >= 2 and "positional" not in k and "probe" not in k:
            sd[k] = v * scale
    g = torch.Generator().manual_seed(seed + 1)
    images = torch.rand(2, 3, cfg.image_size, cfg.image_size, generator=g) * 2 - 1
    return cfg, sd, images


def make_tiny(path, name):
    cfg, sd, images = tiny_case(name)
    taps = {}
    with torch.no_grad():
        out = pe_vit.encode_image(sd, cfg, images, taps)
        emb = pe_vit.l2_normalize(out)
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    np.savez_compressed(path, images=images.numpy(), embedding=emb.numpy(), weight_abs_sum=np.float64(wsum),
                        **{"tap_" + k: v.numpy() for k, v in taps.items()})


def b16_block_case(seed=21):
    import dataclasses
    cfg = dataclasses.replace(reverso_amd.get_config("PE-Core-B16-224"), layers=1)
    sd = weights.synth_weights(cfg, seed=seed, randomize_affine=True)
    for k, v in sd.items():
        if v.dim() >= 2 and "positional" not in k and "probe" not in k:
            sd[k] = v * 2.0
    g = torch.Generator().manual_seed(seed + 1)
    images = torch.rand(1, 3, 224, 224, generator=g) * 2 - 1
    return cfg, sd, images


B16_ROWS = [0, 1, 50, 196]


def make_b16_block(path):
    cfg, sd, images = b16_block_case()
    taps = {}
    with torch.no_grad():
        out = pe_vit.encode_image(sd, cfg, images, taps)
        emb = pe_vit.l2_normalize(out)
    np.savez_compressed(path, rows=np.array(B16_ROWS), ln_pre=taps["ln_pre"][0, B16_ROWS].numpy(),
                        block0=taps["block0"][0, B16_ROWS].numpy(), embedding=emb.numpy(),
                        image_sum=np.float64(images.double().sum().item()))


def search_case(N=4096, D=1024, Q=8, seed=42):
    rng = np.random.default_rng(seed)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    gal[100:120] = gal[100]                       # 20 exact duplicates (tie group)
    perm = rng.permutation(N)[:Q]
    perm[0] = 100                                 # query 0 hits the duplicate group
    qr = gal[perm] + 0.05 * rng.standard_normal((Q, D), dtype=np.float32)
    qr[Q - 1] = rng.standard_normal(D, dtype=np.float32)   # unrelated query: nothing above 0.7
    return gal, qr.astype(np.float32), perm


def make_search(path):
    gal, qr, perm = search_case()
    out = {"perm": perm, "gallery_sum": np.float64(gal.astype(np.float64).sum()),
           "query_sum": np.float64(qr.astype(np.float64).sum())}
    for k in (1, 5, 10, 50):
        for thr in (None, 0.7):
            s, i, c = osearch.search(gal, qr, k, thr)
            tag = f"k{k}_thr{'none' if thr is None else '0p7'}"
            out[tag + "_scores"], out[tag + "_indices"], out[tag + "_counts"] = s, i, c
    # shard-and-merge must equal the unsharded answer
    parts_s, parts_i = [], []
    gn = osearch.normalize_rows(gal)
    qn = osearch.normalize_rows(qr)
    shard = gal.shape[0] // 8
    for p in range(8):
        s, i, c = osearch.search(gn[p * shard:(p + 1) * shard], qn, 10, None, normalize=False)
        parts_s.append(s)
        parts_i.append(np.where(i >= 0, i + p * shard, -1))
    ms, mi, mc = osearch.merge_topk(np.stack(parts_s), np.stack(parts_i), 10, None)
    out["merged_scores"], out["merged_indices"], out["merged_counts"] = ms, mi, mc
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    make_tiny(os.path.join(HERE, "tiny_vit.npz"), "PE-Tiny-T14-56")
    make_tiny(os.path.join(HERE, "tiny_vit_ls.npz"), "PE-Tiny-T14-56-LS")
    make_tiny(os.path.join(HERE, "tiny_vit_n14.npz"), "PE-Tiny-N14-56")      # head_dim 96, no class token (G14 shape family)
    make_b16_block(os.path.join(HERE, "b16_block.npz"))
    make_search(os.path.join(HERE, "search_4096x1024.npz"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
