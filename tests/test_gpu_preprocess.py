"""GPU: device crop + squash-resize (librevo revo_preprocess_crop_resize) is bit-identical to
Pillow's crop(box).resize(BILINEAR) -- the reference's preprocess (core_system.py:200, :439)
-- and to the numpy oracle; then region crops through the embed path."""
import numpy as np
import pytest
import torch
from PIL import Image

import reverso_amd  # noqa: F401
from reverso_amd import _lib, preprocess
from oracle import resize as R

pytestmark = pytest.mark.gpu


def _pil(img, size, box):
    p = Image.fromarray(img)
    if box is not None:
        p = p.crop(box)
    return np.asarray(p.resize((size, size), Image.BILINEAR)).transpose(2, 0, 1)


@pytest.mark.parametrize("h,w,size", [(480, 640, 336), (1080, 1920, 336), (100, 50, 56), (336, 336, 336),
                                      (37, 41, 224), (2160, 3840, 224), (57, 3, 56), (1, 1, 56), (700, 20, 448)])
def test_full_frame_matches_pillow(dev, h, w, size):
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = preprocess.crop_resize_device(torch.from_numpy(img).to(dev), None, size)
    assert out.shape == (1, 3, size, size)
    assert np.array_equal(out[0].cpu().numpy(), _pil(img, size, None))


def test_many_boxes_two_frames_match_pillow_and_oracle(dev):
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (333, 517, 3), dtype=np.uint8), rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)]
    frames = [torch.from_numpy(i).to(dev) for i in imgs]
    boxes = []
    for _ in range(40):
        fi = int(rng.integers(0, 2))
        H, W = imgs[fi].shape[:2]
        x0, y0 = int(rng.integers(0, W - 1)), int(rng.integers(0, H - 1))
        x1, y1 = int(rng.integers(x0 + 1, W + 1)), int(rng.integers(y0 + 1, H + 1))
        boxes.append((fi, x0, y0, x1, y1))
    boxes += [(0, 0, 0, 517, 333), (1, 1279, 719, 1280, 720), (1, 0, 0, 1, 720), (0, 100, 100, 156, 156)]
    out = preprocess.crop_resize_device(frames, boxes, 56).cpu().numpy()
    for i, (fi, x0, y0, x1, y1) in enumerate(boxes):
        want = _pil(imgs[fi], 56, (x0, y0, x1, y1))
        assert np.array_equal(out[i], want), (i, boxes[i])
    # oracle on a few of them
    for i in (0, 7, 41, 43):
        fi, x0, y0, x1, y1 = boxes[i]
        assert np.array_equal(out[i].transpose(1, 2, 0), R.crop_resize_u8(imgs[fi], 56, (x0, y0, x1, y1)))


def test_strided_rows_and_constant_image(dev):
    # a view with padded rows (row_stride > width * 3) and a constant image (must stay constant)
    buf = torch.randint(0, 256, (200, 400, 3), dtype=torch.uint8, device=dev)
    view = buf[:, :300]
    out = preprocess.crop_resize_device(view, [(0, 10, 20, 290, 180)], 112)
    assert np.array_equal(out[0].cpu().numpy(), _pil(view.cpu().numpy(), 112, (10, 20, 290, 180)))
    const = torch.full((97, 131, 3), 77, dtype=torch.uint8, device=dev)
    out = preprocess.crop_resize_device(const, None, 224)
    assert int(out.min()) == 77 and int(out.max()) == 77


def test_bad_boxes_are_rejected(dev):
    img = torch.zeros((50, 60, 3), dtype=torch.uint8, device=dev)
    for box in [(0, 0, 0, 61, 50), (0, 10, 10, 10, 20), (0, -1, 0, 10, 10), (0, 0, 0, 10, 51)]:
        with pytest.raises(_lib.RevoError):
            preprocess.crop_resize_device(img, [box], 56)
    with pytest.raises(_lib.RevoError):
        preprocess.crop_resize_device(torch.zeros((5, 5, 3), dtype=torch.uint8), None, 56)


def test_region_crops_embed_like_host_preprocess(dev):
    """Device crop -> embed == host PIL crop -> embed (same uint8 pixels, so identical vectors)."""
    from reverso_amd.engine import VitEngine
    eng = VitEngine.synthetic("PE-Tiny-T14-56", seed=2, device=0, max_batch=8, randomize_affine=True)
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    boxes = [(0, 0, 0, 320, 240), (0, 20, 30, 200, 220), (0, 150, 10, 310, 100)]
    dev_u8 = preprocess.crop_resize_device(torch.from_numpy(img).to(dev), boxes, 56)
    host_u8 = torch.stack([preprocess.resize_u8(Image.fromarray(img).crop(b[1:]), 56) for b in boxes]).to(dev)
    assert torch.equal(dev_u8, host_u8)
    assert torch.equal(eng.embed(dev_u8), eng.embed(host_u8))
