"""End-to-end ingest (BASELINE.json configs[2]): N JPEGs on disk -> decode -> (detector boxes -> device crops)
-> PE-Core-L14-336 embed -> device gallery -> one search per image, through the SimpleReverso facade.
The detector is synthetic (three fixed boxes per image): GroundedSAM is third-party and out of scope.
    python scripts/ingest_bench.py [n_images]
(1 000 images are 16 batches: the device alone needs 0.44 s for them and the first decode / last embed cannot overlap with
anything, so a run that short tops out near 1 900 images/s whatever the host does; 4 000 images show the steady state)"""
import os, sys, time, tempfile, shutil, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import reverso_amd
from reverso_amd.core_system import SimpleReverso, Regions

if os.environ.get("SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["SWITCH_INTERVAL"]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ONLY = os.environ.get("ONLY_MODE")
root = tempfile.mkdtemp(prefix="ingest_")
folder = os.path.join(root, "images"); os.makedirs(folder)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:480, 0:640]
t0 = time.perf_counter()
for i in range(n):
    base = np.stack([(xx * (i % 7 + 1) + yy) % 256, (yy * 2 + i) % 256, (xx + yy * (i % 5)) % 256], -1).astype(np.float32)
    img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(folder, f"img_{i:05d}.jpg"), quality=90)
print(f"wrote {n} JPEGs (640x480) in {time.perf_counter() - t0:.1f} s", flush=True)

def detector(pil, prompt):
    w, h = pil.size
    return Regions([[0, 0, w // 2, h // 2], [w // 4, h // 4, w - 1, h - 1], [w // 3, 0, w - 1, h // 2]],
                   confidence=[0.9, 0.8, 0.7], class_id=[0, 1, 0], class_names=["person", "car"])

summary = {"what": "scripts/ingest_bench.py: BASELINE.json configs[2] shape end to end on one MI355X through SimpleReverso.create_database "
                   "(JPEGs 640x480 on disk -> decode pool -> pinned staging -> H2D -> PE-Core-L14-336 embed -> device gallery append -> "
                   "delta-shard flush); stage times are wall-clock seconds of the ingest thread (decode_thread_s: summed over the pool's threads)",
           "images": n, "decode_workers": os.environ.get("DECODE_WORKERS", "by mode: 16 host resize, 8 device resize, 6 crops"), "modes": []}
for mode, kw, direct in (("direct PE (one vector per image)", {}, True),
                         ("direct PE, device resize", {"device_resize": True}, True),
                         ("3 detector boxes per image, device crops", {"detector": detector, "region_mode": "crop"}, False)):
    if ONLY and ONLY not in mode:
        continue
    r = SimpleReverso(model_name="PE-Core-L14-336", db_root=os.path.join(root, "db_" + str(len(mode))), max_batch=64,
                      decode_workers=int(os.environ["DECODE_WORKERS"]) if os.environ.get("DECODE_WORKERS") else None, **kw)
    torch.cuda.synchronize()
    if os.environ.get("PROF_CLASSES"):
        from reverso_amd import engine as _eng
        _eng.prof_reset(); _eng.prof_enable(1)
    t0 = time.perf_counter()
    msg = r.create_database(folder, "bench", use_direct_pe=direct)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("PROF_CLASSES"):
        _eng.prof_enable(0)
        rep = _eng.prof_report()
        print("device ms by class:", {k2: round(v["ms"], 1) for k2, v in sorted(rep.items())}, "sum", round(sum(v["ms"] for v in rep.values()), 1))
    vecs = len(r.vector_db.payloads)
    t1 = time.perf_counter()
    r.process_image_direct_pe(os.path.join(folder, "img_00003.jpg"))
    text, items = r.search_similar(0.0, 10)
    dq = time.perf_counter() - t1
    print(f"{mode}: {n} images -> {vecs} vectors in {dt:.2f} s = {n / dt:.1f} images/s ({vecs / dt:.1f} vectors/s); "
          f"one query (decode + embed + top-10) {dq * 1e3:.1f} ms, best hit {items[0]['filename']}", flush=True)
    summary["modes"].append({"mode": mode, "images_per_s": round(n / dt, 1), "vectors_per_s": round(vecs / dt, 1), "vectors": vecs,
                             "seconds": round(dt, 3), "one_query_ms": round(dq * 1e3, 1),
                             "stage_s": {k2: round(v, 3) for k2, v in r.last_ingest_stats.items() if k2.endswith("_s")}})
shutil.rmtree(root, ignore_errors=True)
print(json.dumps(summary))
