"""Perception-Encoder vision-tower variants on the embed path.

The reference pins the model by name only (``core_system.py:177``
``"PE-Core-L14-336"``, fallback to the first available config ``:186/:190``);
the dimensions live in the un-vendored ``perception_models`` package.  The table
below restates them ([UPSTREAM-RECALL]).
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class PEConfig:
    name: str
    image_size: int
    patch_size: int
    width: int
    layers: int
    heads: int
    mlp_dim: int
    out_dim: int
    pool_heads: int = 8
    use_cls: bool = True
    use_ls: bool = False          # LayerScale tensors present in the checkpoint
    ln_eps: float = 1e-5
    rope_theta: float = 10000.0
    # hidden width of the attention-pool head's MLP.  Upstream AttentionPooling takes its own
    # mlp_ratio (4.0), not the tower's: 4 * width.  B16 / L14 have mlp_dim == 4 * width anyway; G14
    # (width 1536, mlp_dim 8960) has a 6144-wide pool MLP.  0 = 4 * width.
    pool_mlp: int = 0

    @property
    def pool_mlp_dim(self) -> int:
        return self.pool_mlp if self.pool_mlp > 0 else 4 * self.width

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def seq(self) -> int:
        return self.grid * self.grid + (1 if self.use_cls else 0)

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def pool_head_dim(self) -> int:
        return self.width // self.pool_heads

    @property
    def patch_k(self) -> int:
        return 3 * self.patch_size * self.patch_size

    def flops_per_image(self) -> float:
        """Algorithmic FLOPs (2*MAC) of one forward, SURVEY.md §8(d) formula."""
        S, W, L, M, D = self.seq, self.width, self.layers, self.mlp_dim, self.out_dim
        G = self.grid * self.grid
        patch = 2.0 * G * self.patch_k * W
        per_layer = 2.0 * S * W * (3 * W) + 2.0 * S * W * W + 4.0 * S * W * M + 4.0 * S * S * W
        pool = 2.0 * S * W * (2 * W) + 2.0 * W * W * 2 + 4.0 * S * W + 4.0 * W * self.pool_mlp_dim
        proj = 2.0 * W * D
        return patch + L * per_layer + pool + proj

    def to_dict(self):
        return asdict(self)


VARIANTS = {
    "PE-Core-B16-224": PEConfig("PE-Core-B16-224", 224, 16, 768, 12, 12, 3072, 1024),
    "PE-Core-L14-336": PEConfig("PE-Core-L14-336", 336, 14, 1024, 24, 16, 4096, 1024),
    "PE-Core-G14-448": PEConfig("PE-Core-G14-448", 448, 14, 1536, 50, 16, 8960, 1280, use_cls=False),
    # Test-only miniature with the same structure (SURVEY.md §8(c) fixture (i)).
    "PE-Tiny-T14-56": PEConfig("PE-Tiny-T14-56", 56, 14, 128, 2, 2, 512, 64, pool_heads=2),
    "PE-Tiny-T14-56-LS": PEConfig("PE-Tiny-T14-56-LS", 56, 14, 128, 2, 2, 512, 64, pool_heads=2, use_ls=True),
    "PE-Tiny-N14-56": PEConfig("PE-Tiny-N14-56", 56, 14, 192, 2, 2, 384, 96, pool_heads=2, use_cls=False),
}

# The model the reference asks for first (core_system.py:177).
DEFAULT_VARIANT = "PE-Core-L14-336"


def get_config(name: str = DEFAULT_VARIANT) -> PEConfig:
    if name not in VARIANTS:
        raise KeyError(f"unknown PE variant {name!r}; available: {sorted(VARIANTS)}")
    return VARIANTS[name]


def available_configs():
    """Mirror of ``pe.CLIP.available_configs()`` (core_system.py:173)."""
    return [n for n in VARIANTS if n.startswith("PE-Core")]
