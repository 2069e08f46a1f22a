"""Host-side image preprocessing of the embed path.

The reference builds ``transforms.get_image_transform(336)`` (``core_system.py:200``)
and applies it per image (``:335``, ``:439``): squash-resize to the model resolution
with bilinear interpolation, ToTensor, Normalize(mean 0.5, std 0.5).  Here the resize
stays on the host (PIL, like the reference) and yields uint8 CHW; ToTensor + Normalize
are fused into the device patchify kernel (``image_dtype = 1`` of ``revo_vit_forward``).

:func:`crop_resize_device` is the device-side variant: decoded
uint8 frames already in HBM are cropped to detector boxes and squash-resized by a HIP kernel
that is bit-identical to PIL's ``crop(box).resize((S, S), BILINEAR)``.
"""
import ctypes as C

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


def crop_resize_device(frames, boxes, size, out=None):
    """Crop + squash-resize on the GPU.

    frames: one uint8 ``[H, W, 3]`` device tensor, or a list of them (sizes may differ).
    boxes:  sequence of ``(frame_index, x0, y0, x1, y1)`` (half-open pixel boxes), or
            ``None`` for one full-frame job per frame.
    Returns uint8 ``[n, 3, size, size]`` on the device, equal to what
    ``Image.fromarray(frame).crop(box).resize((size, size), Image.BILINEAR)`` gives
    (core_system.py:439 applied to the crop of :687-690)."""
    from . import _lib
    if isinstance(frames, torch.Tensor):
        frames = [frames]
    if not frames:
        raise ValueError("crop_resize_device: no frames")
    dev = frames[0].device
    for f in frames:
        if not f.is_cuda:
            raise _lib.RevoError("crop_resize_device: frames must be device tensors (no CPU fallback)")
        if f.dtype != torch.uint8 or f.dim() != 3 or f.shape[2] != 3 or f.stride(2) != 1 or f.stride(1) != 3:
            raise ValueError("crop_resize_device: frames must be uint8 [H, W, 3] with packed RGB pixels")
        if f.device != dev:
            raise ValueError("crop_resize_device: all frames must live on one device")
    if boxes is None:
        boxes = [(i, 0, 0, f.shape[1], f.shape[0]) for i, f in enumerate(frames)]
    n = len(boxes)
    if out is None:
        out = torch.empty((n, 3, size, size), dtype=torch.uint8, device=dev)
    if n == 0:
        return out
    jobs = (_lib.CropJob * n)()
    for i, (fi, x0, y0, x1, y1) in enumerate(boxes):
        f = frames[int(fi)]
        j = jobs[i]
        j.src, j.height, j.width, j.row_stride = f.data_ptr(), f.shape[0], f.shape[1], f.stride(0)
        j.x0, j.y0, j.x1, j.y1 = int(x0), int(y0), int(x1), int(y1)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.revo_preprocess_crop_resize(jobs, n, int(size), _lib.ptr(out), _lib.current_stream()),
                   "revo_preprocess_crop_resize")
    return out


def clamp_box(box, width, height):
    """xyxy (floats or ints) -> half-open integer box inside the image, at least one pixel
    on each side (floor of the low corner, ceil of the high corner)."""
    bx0, by0, bx1, by1 = [float(v) for v in box]
    x0, x1 = int(np.floor(min(bx0, bx1))), int(np.ceil(max(bx0, bx1)))
    y0, y1 = int(np.floor(min(by0, by1))), int(np.ceil(max(by0, by1)))
    x0, y0 = max(0, min(x0, width - 1)), max(0, min(y0, height - 1))
    x1, y1 = max(x0 + 1, min(x1, width)), max(y0 + 1, min(y1, height))
    return x0, y0, x1, y1
