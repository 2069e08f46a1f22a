"""Host-side image preprocessing of the embed path.

The reference builds ``transforms.get_image_transform(336)`` (``core_system.py:200``)
and applies it per image (``:335``, ``:439``): squash-resize to the model resolution
with bilinear interpolation, ToTensor, Normalize(mean 0.5, std 0.5).  Here the resize
stays on the host (PIL, like the reference) and yields uint8 CHW; ToTensor + Normalize
are fused into the device patchify kernel (``image_dtype = 1`` of ``revo_vit_forward``).
"""
import numpy as np
import torch
from PIL import Image


def to_pil(image):
    """np.ndarray | path | PIL -> PIL RGB (core_system.py:435-439)."""
    if isinstance(image, np.ndarray):
        image = Image.fromarray(image)
    elif isinstance(image, (str, bytes)) or hasattr(image, "__fspath__"):
        image = Image.open(image)
    return image.convert("RGB")


def resize_u8(image, size):
    """PIL RGB -> uint8 tensor [3, size, size] (bilinear squash, no crop)."""
    im = to_pil(image)
    if im.size != (size, size):
        im = im.resize((size, size), Image.BILINEAR)
    arr = np.asarray(im, dtype=np.uint8)            # HWC
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


def batch_u8(images, size, pin=False):
    """list of images -> uint8 [B, 3, size, size] host tensor."""
    out = torch.empty((len(images), 3, size, size), dtype=torch.uint8)
    if pin:
        out = out.pin_memory()
    for i, im in enumerate(images):
        out[i] = resize_u8(im, size)
    return out


def normalize_u8(u8):
    """What the device kernel computes from uint8: (v/255 - 0.5)/0.5."""
    return (u8.to(torch.float32) / 255.0 - 0.5) / 0.5
