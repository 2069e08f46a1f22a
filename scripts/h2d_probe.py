"""Host-side costs of the ingest's upload path on this box: writing into pinned memory, asynchronous H2D copies
(whole tensor / slice / many small), the device crop + resize call.  python scripts/h2d_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import reverso_amd
from reverso_amd import preprocess as pp
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - t0) / n * 1e3
MB = 1 << 20
pin = torch.empty(96 * MB, dtype=torch.uint8).pin_memory()
pag = torch.empty(96 * MB, dtype=torch.uint8)
arr = np.random.randint(0, 255, (480, 640, 3), dtype=np.uint8)
print("is_pinned", pin.is_pinned(), pin[:59 * MB].is_pinned())
def wr(dst):
    for j in range(64):
        np.copyto(dst[j * MB:j * MB + arr.size].view(arr.shape).numpy(), arr)
print("64 frames (59 MB) numpy -> pinned   host ms %.2f" % t(lambda: wr(pin))[0])
print("64 frames (59 MB) numpy -> pageable host ms %.2f" % t(lambda: wr(pag))[0])
print("H2D 59 MB pinned slice non_blocking: host %.2f ms, done %.2f ms" % t(lambda: pin[:59 * MB].to(dev, non_blocking=True)))
print("H2D 96 MB pinned whole non_blocking: host %.2f ms, done %.2f ms" % t(lambda: pin.to(dev, non_blocking=True)))
print("H2D 59 MB pageable:                  host %.2f ms, done %.2f ms" % t(lambda: pag[:59 * MB].to(dev, non_blocking=True)))
print("H2D 64 x 0.9 MB pinned slices:       host %.2f ms, done %.2f ms" % t(lambda: [pin[j * MB:j * MB + arr.size].to(dev, non_blocking=True) for j in range(64)]))
d = pin[:64 * MB].to(dev)
frames = [d[j * MB:j * MB + arr.size].view(480, 640, 3) for j in range(64)]
print("crop_resize_device 64 frames -> 336: host %.2f ms, done %.2f ms" % t(lambda: pp.crop_resize_device(frames, None, 336)))
stage = torch.zeros((64, 3, 336, 336), dtype=torch.uint8).pin_memory()
print("H2D 21.7 MB staged batch:            host %.2f ms, done %.2f ms" % t(lambda: stage.to(dev, non_blocking=True)))
# the crop-mode batch of scripts/ingest_bench.py: 64 frames, three boxes each -> 192 crops of 336 x 336
w, h = 640, 480
boxes = []
for i in range(64):
    boxes += [(i, 0, 0, w // 2, h // 2), (i, w // 4, h // 4, w - 1, h - 1), (i, w // 3, 0, w - 1, h // 2)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pp.crop_resize_device(frames, boxes, 336); torch.cuda.synchronize()
e0.record()
for _ in range(10): pp.crop_resize_device(frames, boxes, 336)
e1.record(); torch.cuda.synchronize()
print("crop_resize_device 192 crops of 64 frames: device %.3f ms per call" % (e0.elapsed_time(e1) / 10))
t0 = time.perf_counter()
for _ in range(10): pp.crop_resize_device(frames, boxes, 336)
print("   host %.3f ms per call" % ((time.perf_counter() - t0) / 10 * 1e3)); torch.cuda.synchronize()
