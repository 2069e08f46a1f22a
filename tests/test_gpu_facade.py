"""End to end through the SimpleReverso façade: BASELINE.json configs[0]
(PE-Core-B16-224 on 32 local JPEGs, brute-force cosine top-5) against the CPU oracle
run on the same decoded pixels; persistence, resume and the reference's result
formatting."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image, ImageDraw

import reverso_amd
from reverso_amd import preprocess as pp
from reverso_amd import weights
from reverso_amd.core_system import SimpleReverso
from oracle import pe_vit, search as osearch

pytestmark = pytest.mark.gpu


def _make_jpegs(folder, n=32, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)
    paths = []
    for i in range(n):
        w, h = int(rng.integers(180, 400)), int(rng.integers(160, 360))
        base = rng.integers(0, 256, (3,))
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([(base[c] + (xx * rng.integers(1, 4) + yy * rng.integers(1, 4)) // 2) % 256 for c in range(3)], -1)
        im = Image.fromarray(arr.astype(np.uint8))
        d = ImageDraw.Draw(im)
        for _ in range(4):
            x0, y0 = int(rng.integers(0, w - 20)), int(rng.integers(0, h - 20))
            d.ellipse([x0, y0, x0 + int(rng.integers(10, 120)), y0 + int(rng.integers(10, 120))],
                      fill=tuple(int(v) for v in rng.integers(0, 256, 3)))
        p = os.path.join(folder, f"img_{i:03d}.jpg")
        im.save(p, quality=90)
        paths.append(p)
    return paths


@pytest.fixture(scope="module")
def system(tmp_path_factory, dev):
    root = tmp_path_factory.mktemp("facade")
    folder = str(root / "images")
    paths = _make_jpegs(folder)
    (root / "images" / "broken.jpg").write_bytes(b"not a jpeg")
    r = SimpleReverso(model_name="PE-Core-B16-224", db_root=str(root / "simple_reverso_db"), max_batch=8)
    return r, folder, paths, root


def test_config1_b16_32_jpegs_top5(system):
    r, folder, paths, root = system
    seen = []
    msg = r.create_database(folder, "cfg1", use_direct_pe=True, progress_callback=lambda m, v=None: seen.append((m, v)))
    assert "✅ Successfully processed: 32 images" in msg and "⚠️ Failed to process: 1 images" in msg
    assert "❌ Error processing broken.jpg" in msg and "🎯 Database 'cfg1' ready for searching!" in msg
    assert seen[-1][1] == 1.0 and r.current_database == "simple_reverso_cfg1" and len(r.vector_db) == 32
    assert r.list_databases() == ["cfg1"]

    # oracle on the same decoded + resized pixels, fp32, one image per forward like the reference
    cfg = reverso_amd.get_config("PE-Core-B16-224")
    sd = weights.synth_weights(cfg, seed=0)
    files = sorted(paths)
    u8 = torch.stack([pp.resize_u8(p, 224) for p in files])
    ref = pe_vit.embed_batch1(sd, cfg, pe_vit.preprocess_u8(u8)).numpy()
    stored = r.vector_db.gallery.read().cpu().numpy()
    order = [r.vector_db.payloads[i]["filename"] for i in range(32)]
    assert order == [os.path.basename(p) for p in files]
    assert ((stored * ref).sum(-1) >= 0.9999).all()

    # query = one gallery image, reference flow: process_image_direct_pe then search_similar
    q = 7
    embs, metas = r.process_image_direct_pe(files[q])
    assert metas[0]["detected_class"] == "full_image" and metas[0]["bbox"][:2] == [0, 0]
    text, items = r.search_similar(similarity_threshold=0.0, max_results=5)
    assert text.startswith("🎯 Found 5 similar regions:") and len(items) == 5
    assert items[0]["filename"] == os.path.basename(files[q]) and abs(items[0]["score"] - 1.0) < 1e-5
    assert items[0]["image"] is not None and max(items[0]["image"].size) <= 400
    assert all(items[i]["score"] >= items[i + 1]["score"] for i in range(4))
    # scores agree with the fp32 oracle for the same files to 1e-3
    rs, ri, rc = osearch.search(ref, ref[q:q + 1], 32)
    oracle_score = {order[int(i)]: float(s) for s, i in zip(rs[0], ri[0])}
    for it in items:
        assert abs(it["score"] - oracle_score[it["filename"]]) <= 1e-3
    # the product's own search over its stored vectors is exact against the oracle over the same vectors
    hits = r.vector_db.search(embs[0], limit=5)
    s2, i2, _ = osearch.search(stored, embs[0].numpy()[None], 5)
    assert [h.payload["filename"] for h in hits] == [order[int(i)] for i in i2[0]]
    # threshold above every non-identical score: only the image itself
    text, items = r.search_similar(similarity_threshold=0.99999, max_results=5)
    assert len(items) >= 1 and items[0]["filename"] == os.path.basename(files[q])
    text, items = r.search_similar(similarity_threshold=1.5, max_results=5)
    assert items == [] and text == "❌ No similar regions found above threshold 1.5"


def test_persistence_and_reload(system):
    r, folder, paths, root = system
    before = r.vector_db.gallery.read().cpu()
    assert r.load_database("cfg1") == "✅ Loaded database: cfg1"
    after = r.vector_db.gallery.read().cpu()
    assert torch.equal(before, after) and len(r.vector_db.payloads) == 32
    r.process_image_direct_pe(sorted(paths)[3])
    _, items = r.search_similar(0.0, 3)
    assert items[0]["filename"] == "img_003.jpg"


def test_detector_mode_and_resume(system):
    r, folder, paths, root = system
    # no detector injected: one full-frame region per image, same vectors as direct PE
    n = r.detect_regions(paths[0], "person . car")
    assert n == 1
    embs, metas = r.extract_embeddings(paths[0])
    assert len(embs) == 1 and metas[0]["mask_status"] == "missing_or_unavailable"
    d_embs, _ = r.process_image_direct_pe(paths[0])
    assert torch.equal(embs[0], d_embs[0])
    # stop after the first batch, then resume: the final gallery equals an uninterrupted build
    calls = {"n": 0}

    def cb(m, v=None):
        calls["n"] += 1
        if m.startswith("🔄 Processing 9/"):
            r.request_stop()
    msg = r.create_database(folder, "resumed", use_direct_pe=True, progress_callback=cb)
    assert "⏸️ Processing stopped" in msg
    msg = r.create_database(folder, "resumed", use_direct_pe=True, resume_from_checkpoint=True)
    assert "📋 Resuming from checkpoint" in msg and "🧹 Cleaned up checkpoint file" in msg
    a = r.vector_db.gallery.read().cpu()
    names = [p["filename"] for p in r.vector_db.payloads]
    assert sorted(names) == sorted(os.path.basename(p) for p in paths) and len(names) == 32
    assert r.load_database("cfg1").startswith("✅")
    b = r.vector_db.gallery.read().cpu()
    ref_names = [p["filename"] for p in r.vector_db.payloads]
    perm = [ref_names.index(nm) for nm in names]
    assert torch.equal(a, b[perm])
    assert r.create_database(str(root), "empty", use_direct_pe=True).endswith(f"❌ No images found in {root}")


def test_region_crop_mode_embeds_each_box(tmp_path, dev):
    """region_mode="crop" (SURVEY §8(f) row 3): every detector box gets the embedding of its own
    crop -- equal to embedding PIL's crop().resize() of that box -- and the gallery built from
    a folder stores one distinct vector per region."""
    from PIL import Image
    from reverso_amd import preprocess as pp
    from reverso_amd.core_system import Regions
    folder = str(tmp_path / "images")
    paths = _make_jpegs(folder, n=3, seed=4)

    def detector(pil, prompt):
        w, h = pil.size
        masks = np.zeros((2, h, w), dtype=bool)
        masks[0, 10:h // 2, 5:w // 2] = True
        masks[1, h // 3:h - 7, w // 4:w - 3] = True
        return Regions([[5, 10, w // 2, h // 2], [w // 4, h // 3, w - 3, h - 7]], mask=masks, confidence=[0.9, 0.8],
                       class_id=[0, 1], class_names=["person", "car"])

    r = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=str(tmp_path / "db"), max_batch=8, detector=detector,
                      region_mode="crop")
    assert r.detect_regions(paths[0], "person . car") == 2
    embs, metas = r.extract_embeddings(paths[0])
    assert len(embs) == 2 and metas[0]["detected_class"] == "person" and metas[1]["detected_class"] == "car"
    pil = Image.open(paths[0]).convert("RGB")
    for e, m in zip(embs, metas):
        x0, y0, x1, y1 = m["bbox"]                         # inclusive, mask derived
        crop = pil.crop((x0, y0, x1 + 1, y1 + 1))
        want = r.pe_model.embed(pp.resize_u8(crop, 56)[None].to(dev))[0].cpu()
        assert torch.equal(e, want)
    assert not torch.equal(embs[0], embs[1])
    msg = r.create_database(folder, "regions", text_prompt="person . car")
    assert "✅" in msg and len(r.vector_db.payloads) == 6
    g = r.vector_db.gallery.read().cpu()
    assert torch.unique(g, dim=0).shape[0] == 6
    # the first region of image 0 finds itself
    r.detect_regions(paths[0], "person . car")
    r.extract_embeddings(paths[0])
    _, items = r.search_similar(0.0, 1)
    assert items[0]["filename"] == os.path.basename(paths[0]) and items[0]["score"] > 0.9999
    assert items[0]["bbox"] == metas[0]["bbox"]


def test_pipelined_ingest_equals_one_batch(tmp_path, dev):
    """create_database overlaps decode, embed and bookkeeping across three batches (staging and result buffers used
    alternately, region vectors stored one batch late).  Many small batches must store exactly what one big batch
    stores, in the same order, in every mode -- a broken file in the middle included."""
    from reverso_amd.core_system import Regions
    folder = str(tmp_path / "images")
    paths = _make_jpegs(folder, n=21, seed=9)
    (tmp_path / "images" / "img_010a_broken.jpg").write_bytes(b"not a jpeg")
    # four of the files in other modes (core_system.py:439 converts whatever it opens to RGB): grey, RGBA and palette PNGs,
    # a CMYK JPEG -- same names, so that the order of the collection does not change
    for j, mode in ((2, "L"), (5, "RGBA"), (8, "P"), (13, "CMYK")):
        im = Image.open(paths[j]).convert(mode)
        os.remove(paths[j])
        im.save(paths[j][:-4] + (".jpg" if mode == "CMYK" else ".png"))

    def detector(pil, prompt):
        w, h = pil.size
        return Regions([[0, 0, w // 2, h // 2], [w // 4, h // 4, w - 1, h - 1], [w // 3, 0, w - 1, h // 2]],
                       confidence=[0.9, 0.8, 0.7], class_id=[0, 1, 0], class_names=["person", "car"])

    direct_runs = []
    for kw, direct in (({}, True), ({"device_resize": True}, True), ({"detector": detector, "region_mode": "crop"}, False),
                       ({"detector": detector}, False)):
        got = []
        for mb in (4, 64):
            r = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=str(tmp_path / f"db{len(got)}_{mb}_{len(kw)}_{direct}"),
                              max_batch=mb, **kw)
            msg = r.create_database(folder, "p", use_direct_pe=direct)
            assert "✅" in msg and "❌ Error processing img_010a_broken.jpg" in msg
            got.append((r.vector_db.gallery.read().cpu(), [(p["filename"], p["bbox"]) for p in r.vector_db.payloads]))
        assert got[0][1] == got[1][1]
        assert len(got[0][1]) == (21 if direct else 63)
        assert torch.equal(got[0][0], got[1][0])
        if direct:
            direct_runs.append(got[0])
    # host resize and device resize store the same vectors, the odd image modes included
    assert direct_runs[0][1] == direct_runs[1][1] and torch.equal(direct_runs[0][0], direct_runs[1][0])


def test_device_resize_ingest_equals_host_resize(system):
    """device_resize=True builds the same gallery, bit for bit."""
    r, folder, paths, root = system
    assert r.load_database("cfg1").startswith("✅")
    host = r.vector_db.gallery.read().cpu()
    host_names = [p["filename"] for p in r.vector_db.payloads]
    r.device_resize = True
    try:
        msg = r.create_database(folder, "devresize", use_direct_pe=True)
    finally:
        r.device_resize = False
    assert "✅" in msg
    names = [p["filename"] for p in r.vector_db.payloads]
    perm = [host_names.index(nm) for nm in names]
    assert torch.equal(r.vector_db.gallery.read().cpu(), host[perm])


def test_resume_after_stop_in_storage_phase_builds_the_database(system, tmp_path):
    """A stop (or crash) after the last embed batch but before the collection was completed leaves a build directory whose
    shards cover every file, and no database.  Resuming must complete it from those shards instead of answering "already
    complete", and a later create_database for another name must not inherit those vectors."""
    r, folder, paths, root = system
    sub = tmp_path / "imgs"
    sub.mkdir()
    for p in paths[:6]:
        (sub / os.path.basename(p)).write_bytes(open(p, "rb").read())

    def stop_at_storage(msg, v=None):
        if msg.startswith("📦 Recreated collection"):
            r.request_stop()
    out = r.create_database(str(sub), "stopped", use_direct_pe=True, progress_callback=stop_at_storage)
    assert "Processing stopped" in out
    assert not os.path.isdir(os.path.join(r.db_root, "stopped"))                      # nothing is visible as a database yet
    assert os.path.exists(os.path.join(r.db_root, "stopped.building", "manifest.jsonl")) and "stopped.building" not in r.list_databases()
    assert r._partial_embeddings == [] and r._partial_metadata == [] and r._build_store is None
    out = r.create_database(str(sub), "stopped", use_direct_pe=True, resume_from_checkpoint=True)
    assert "All files already embedded" in out and "ready for searching" in out, out
    assert os.path.exists(os.path.join(r.db_root, "stopped", "manifest.jsonl")) and len(r.vector_db) == 6
    assert not os.path.isdir(os.path.join(r.db_root, "stopped.building"))
    # complete now: a second resume has nothing to do
    out = r.create_database(str(sub), "stopped", use_direct_pe=True, resume_from_checkpoint=True)
    assert "ready for searching" in out or "already processed" in out
    # no leak into the next collection
    other = tmp_path / "other"
    other.mkdir()
    (other / "a.jpg").write_bytes(open(paths[10], "rb").read())
    out = r.create_database(str(other), "fresh", use_direct_pe=True)
    assert "Total embeddings stored: 1" in out and len(r.vector_db) == 1


def test_killed_build_resumes_from_its_delta_shards(tmp_path, dev):
    """Persistence that scales: a build writes its collection as delta shards (one .npy + one manifest
    line per checkpoint, never the whole set).  A build of 6 batches is killed after its third shard -- an exception out
    of the progress callback, nothing gets to clean up; a stray, half-written shard file is left behind as a crash would
    -- and a NEW process-equivalent (fresh SimpleReverso) resumes it: the finished gallery holds exactly the bytes, ids
    aside, of an uninterrupted build, in the same order; what was on disk before the kill was not rewritten."""
    folder = str(tmp_path / "images")
    paths = _make_jpegs(folder, n=24, seed=21)
    ref = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=str(tmp_path / "db_ref"), max_batch=4)
    assert "✅" in ref.create_database(folder, "g", use_direct_pe=True)
    want = ref.vector_db.gallery.read().cpu()
    want_names = [p["filename"] for p in ref.vector_db.payloads]

    class Kill(BaseException):
        pass

    def killer(msg, v=None):
        if msg.startswith("💾 Checkpoint: shard 2 "):
            raise Kill()
    root = str(tmp_path / "db")
    r = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4, checkpoint_interval_s=0.0)
    with pytest.raises(Kill):
        r.create_database(folder, "g", use_direct_pe=True, progress_callback=killer)
    build = os.path.join(root, "g.building")
    lines = [json.loads(x) for x in open(os.path.join(build, "manifest.jsonl")).read().splitlines()]
    assert [x.get("shard") for x in lines[1:]] == [0, 1, 2] and sum(x["rows"] for x in lines[1:]) == 12
    before = {f: os.path.getmtime(os.path.join(build, f)) for f in os.listdir(build) if f.startswith("vectors.")}
    assert sorted(before) == ["vectors.00000.f32.npy", "vectors.00001.f32.npy", "vectors.00002.f32.npy"]
    open(os.path.join(build, "vectors.00003.f32.npy.tmp.npy"), "wb").write(b"torn")         # what a crash mid-write leaves
    open(os.path.join(build, "manifest.jsonl"), "a").write('{"shard": 3, "file": "vectors.0')  # and a torn manifest line
    del r
    r2 = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4, checkpoint_interval_s=0.0)
    assert "g" not in r2.list_databases() and r2.load_database("g").startswith("❌")
    msg = r2.create_database(folder, "g", use_direct_pe=True, resume_from_checkpoint=True)
    assert "📋 Resuming from checkpoint: 12 files already processed" in msg and "ready for searching" in msg, msg[-600:]
    assert "🔄 Processing 12/12" in msg and "Processing 13/" not in msg                    # only the other 12 files were embedded
    got = r2.vector_db.gallery.read().cpu()
    assert [p["filename"] for p in r2.vector_db.payloads] == want_names
    assert torch.equal(got, want)
    done = os.path.join(root, "g")
    for f, t in before.items():
        assert os.path.getmtime(os.path.join(done, f)) == t                                # the first three shards were not rewritten
    # reload from disk: the same bytes; and the store keeps appending deltas after a save
    r3 = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4)
    assert r3.load_database("g").startswith("✅")
    assert torch.equal(r3.vector_db.gallery.read().cpu(), want) and len(r3.vector_db.payloads) == 24
    n_files = len(os.listdir(done))
    r3.vector_db.upsert(want[:2], ["x", "y"], [{"filename": "x"}, {"filename": "y"}])
    r3.vector_db.save()
    assert len(os.listdir(done)) == n_files + 1
    r4 = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4)
    assert r4.load_database("g").startswith("✅") and len(r4.vector_db) == 26


def test_load_failure_falls_back_to_the_first_available_config(tmp_path, dev, capsys):
    """core_system.py:183-191: if the target model fails to LOAD, the reference falls back to available_configs()[0].
    Here a checkpoint file that does not fit PE-Core-L14-336 (it is a PE-Core-B16-224 one) makes the target's load fail
    -- by name, in check_state_dict -- and the fallback config, which it does fit, is what the facade ends up with."""
    from safetensors.torch import save_file
    from reverso_amd import engine
    cfg = reverso_amd.get_config("PE-Core-B16-224")
    assert reverso_amd.available_configs()[0] == cfg.name
    sd = weights.synth_weights(cfg, seed=5)
    ck = str(tmp_path / "b16.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, ck)
    r = SimpleReverso(model_name="PE-Core-L14-336", checkpoint=ck, db_root=str(tmp_path / "db"), max_batch=2)
    out = capsys.readouterr().out
    assert "❌ Failed to load PE-Core-L14-336" in out and "lacks" in out and "🔄 Using fallback: PE-Core-B16-224" in out
    assert r.pe_model.cfg.name == cfg.name
    g = torch.Generator().manual_seed(1)
    u8 = torch.randint(0, 256, (2, 3, 224, 224), generator=g, dtype=torch.uint8)
    eng = engine.VitEngine(cfg, {k: v.to(dev) for k, v in sd.items()}, device=0, max_batch=2)
    assert torch.equal(r.pe_model.embed(u8.to(dev)), eng.embed(u8.to(dev)))
    eng.close()
    # a checkpoint that fits nothing: the fallback's own failure propagates (the reference's second from_config is not guarded)
    bad = dict(sd)
    bad["visual.foo"] = torch.zeros(3)
    ck2 = str(tmp_path / "bad.safetensors")
    save_file({k: v.contiguous() for k, v in bad.items()}, ck2)
    with pytest.raises(KeyError, match=r"visual\.foo"):
        SimpleReverso(model_name="PE-Core-B16-224", checkpoint=ck2, db_root=str(tmp_path / "db2"), max_batch=2)


def test_resume_refuses_a_build_of_another_folder_model_or_mode(tmp_path, dev):
    """A stale <db>.building of ANOTHER source folder (or model, or mode) must not be continued: its rows would be mixed
    with this build's.  The manifest header records what a build is made from; a resume on a mismatch says why and
    starts fresh.  delete_database takes an unfinished build of that name and its checkpoint note with it."""
    fa, fb = str(tmp_path / "a"), str(tmp_path / "b")
    _make_jpegs(fa, n=8, seed=1)
    _make_jpegs(fb, n=8, seed=2)
    root = str(tmp_path / "db")

    class Kill(BaseException):
        pass

    def killer(msg, v=None):
        if msg.startswith("💾 Checkpoint: shard 0 "):
            raise Kill()
    r = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4, checkpoint_interval_s=0.0)
    with pytest.raises(Kill):
        r.create_database(fa, "g", use_direct_pe=True, progress_callback=killer)
    build = os.path.join(root, "g.building")
    hdr = json.loads(open(os.path.join(build, "manifest.jsonl")).readline())
    assert hdr["build"]["folder_path"] == os.path.abspath(fa) and hdr["build"]["model"] == "PE-Tiny-T14-56"
    assert hdr["build"]["use_direct_pe"] is True and hdr["build"]["region_mode"] == "global"
    r2 = SimpleReverso(model_name="PE-Tiny-T14-56", db_root=root, max_batch=4)
    msg = r2.create_database(fb, "g", use_direct_pe=True, resume_from_checkpoint=True)            # another folder
    assert "different folder_path" in msg and "Starting fresh" in msg and "ready for searching" in msg, msg[-500:]
    assert len(r2.vector_db) == 8 and all(fb in p["image_source"] for p in r2.vector_db.payloads)
    # the same folder in another mode is refused too
    with pytest.raises(Kill):
        r.create_database(fa, "h", use_direct_pe=True, progress_callback=killer)
    msg = r2.create_database(fa, "h", use_direct_pe=False, resume_from_checkpoint=True)
    assert "different use_direct_pe" in msg and "Starting fresh" in msg
    # delete: the database, a stale build of that name and the checkpoint note
    with pytest.raises(Kill):
        r.create_database(fa, "h", use_direct_pe=True, progress_callback=killer)
    assert os.path.isdir(os.path.join(root, "h.building"))
    assert r2.delete_database("h").startswith("✅")
    assert not os.path.exists(os.path.join(root, "h")) and not os.path.exists(os.path.join(root, "h.building"))
    assert not os.path.exists(os.path.join(root, "checkpoints", "h_checkpoint.json"))


def test_two_threads_share_one_instance(tmp_path, dev):
    """The reference's UI drives ONE global SimpleReverso from Gradio's worker threads with no locks (ui.py:19-20): a build
    runs in one callback (ui.py:86-94) while another embeds a query and searches the loaded database (ui.py:39, :131-142,
    swapping `region_embeddings` in place, :128-133) and a third sets the stop flag the build polls (ui.py:109,
    core_system.py:457-459, :542-545).  Here: thread A builds a 256-image database; thread B, meanwhile, embeds + searches
    against a previously loaded database and finally asks A to stop.  B's answers equal the single-threaded ones bit for
    bit (the engine's workspace and the handles are serialised by the instance's lock), A returns the reference's stop
    message, the loaded database stays the searchable one, and the resumed build ends with the gallery bytes an
    uninterrupted build gives."""
    import threading
    import time
    folder_small = str(tmp_path / "small")
    small = _make_jpegs(folder_small, n=24, seed=3)
    folder_big = str(tmp_path / "big")
    _make_jpegs(folder_big, n=256, seed=4)
    r = SimpleReverso(model_name="PE-Core-B16-224", db_root=str(tmp_path / "db"), max_batch=8, checkpoint_interval_s=0.05)
    assert "ready for searching" in r.create_database(folder_small, "loaded", use_direct_pe=True)
    # single-threaded answers first
    want = []
    for p in small[:6]:
        emb, _ = r.process_image_direct_pe(p)
        text, items = r.search_similar(similarity_threshold=0.0, max_results=5)
        want.append((emb[0].clone(), text, [(it["filename"], it["score"], it["bbox"]) for it in items]))
    assert all(w[2][0][0] == os.path.basename(p) for w, p in zip(want, small[:6]))           # each image finds itself first

    out, errors, started = {}, [], threading.Event()

    def slow_progress(message, value=None):
        """A UI's progress callback (ui.py:86-94 hands one in): it also keeps this build going long enough for the other
        thread's queries -- 256 small JPEGs are a fraction of a second of device time."""
        started.set()
        time.sleep(0.02)

    def build():
        try:
            out["msg"] = r.create_database(folder_big, "big", use_direct_pe=True, progress_callback=slow_progress)
        except Exception as e:                       # the reference never raises to the UI (core_system.py:585-591)
            errors.append(e)

    def query():
        try:
            assert started.wait(60)
            rounds = 0
            t_end = time.time() + 120
            while rounds < 3 and time.time() < t_end:
                for p, (e0, text0, items0) in zip(small[:6], want):
                    emb, _ = r.process_image_direct_pe(p)
                    text, items = r.search_similar(similarity_threshold=0.0, max_results=5)
                    assert torch.equal(emb[0], e0), "a query embedded while a build runs differs"
                    assert text == text0 and [(it["filename"], it["score"], it["bbox"]) for it in items] == items0
                    assert r.current_database == "simple_reverso_loaded"              # the build has not replaced it
                rounds += 1
            out["rounds"] = rounds
            out["still_building"] = r._is_building("big") and ta.is_alive()      # the queries did overlap the build
            r.request_stop()                         # ui.py:109
        except Exception as e:
            errors.append(e)
            r.request_stop()

    ta, tb = threading.Thread(target=build), threading.Thread(target=query)
    ta.start(); tb.start()
    tb.join(180); ta.join(180)
    assert not ta.is_alive() and not tb.is_alive()
    assert not errors, errors
    assert out["rounds"] == 3 and out["still_building"], out
    assert "⏸️ Processing stopped. You can resume later." in out["msg"], out["msg"][-300:]
    assert r.list_databases() == ["loaded"] and len(r.vector_db) == 24               # the half-built one is not a database
    # resume (core_system.py:524-538) and compare with an uninterrupted build of the same folder
    msg = r.create_database(folder_big, "big", use_direct_pe=True, resume_from_checkpoint=True)
    assert "ready for searching" in msg and len(r.vector_db) == 256, msg[-300:]
    resumed = r.vector_db.gallery.read(0, 256).cpu()
    names = [pl["filename"] for pl in r.vector_db.payloads]
    r2 = SimpleReverso(model_name="PE-Core-B16-224", db_root=str(tmp_path / "db2"), max_batch=8)
    assert "ready for searching" in r2.create_database(folder_big, "big", use_direct_pe=True)
    assert [pl["filename"] for pl in r2.vector_db.payloads] == names
    assert torch.equal(r2.vector_db.gallery.read(0, 256).cpu(), resumed)
