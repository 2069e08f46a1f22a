"""What the decode pool alone delivers on this box (16 threads, 640x480 JPEGs): decode only / + numpy view / + copy into a
pinned slab / + PIL resize to 336 (the host-resize path).   python scripts/decode_probe.py [n]"""
import os, sys, time, tempfile, shutil, threading
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
from PIL import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
root = tempfile.mkdtemp(prefix="dec_")
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:480, 0:640]
for i in range(n):
    base = np.stack([(xx * (i % 7 + 1) + yy) % 256, (yy * 2 + i) % 256, (xx + yy * (i % 5)) % 256], -1).astype(np.float32)
    Image.fromarray(np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)).save(os.path.join(root, f"{i:05d}.jpg"), quality=90)
paths = sorted(os.path.join(root, f) for f in os.listdir(root))
pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
def dec(p): Image.open(p).convert("RGB")
def dec_np(p): np.asarray(Image.open(p).convert("RGB"))
def dec_pin(p, j=[0]):
    a = np.asarray(Image.open(p).convert("RGB"))
    np.copyto(pin[:a.size].view(a.shape).numpy(), a)
def dec_bytes(p):
    im = Image.open(p).convert("RGB")
    b = im.tobytes()
def dec_resize(p): Image.open(p).convert("RGB").resize((336, 336), Image.BILINEAR)
for workers in (16, 32):
    pool = ThreadPoolExecutor(max_workers=workers)
    for name, fn in (("decode", dec), ("decode+asarray", dec_np), ("decode+tobytes", dec_bytes), ("decode+asarray+pinned copy", dec_pin), ("decode+resize336", dec_resize)):
        list(pool.map(fn, paths[:64]))
        t0 = time.perf_counter(); list(pool.map(fn, paths)); dt = time.perf_counter() - t0
        print(f"{workers} threads  {name:28s} {n / dt:8.0f} images/s", flush=True)
    pool.shutdown()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
shutil.rmtree(root, ignore_errors=True)
