"""CPU: host-side logic above the C ABI (no device calls): database admin strings,
checkpoint round trip, region metadata, preprocessing, the sharded-search protocol
on gloo with world_size 2 (the oracle is injected as the compute step)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import preprocess as pp
from reverso_amd import store as st
from oracle import search as osearch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bare_facade(tmp_path):
    from reverso_amd.core_system import SimpleReverso
    import threading
    r = object.__new__(SimpleReverso)            # no device: only the host-side methods are exercised
    r.db_root = str(tmp_path / "simple_reverso_db")
    r._lock = threading.RLock()
    r.vector_db = None
    r.current_database = None
    r.region_embeddings = None
    r._stop_requested = False
    return r


def test_database_admin_strings(tmp_path):
    r = _bare_facade(tmp_path)
    assert r.list_databases() == []
    assert r.load_database("") == "❌ Please provide a database name"
    assert r.load_database("nope") == "❌ Database not found: nope"
    assert r.delete_database("nope") == "❌ Database not found: nope"
    assert r.unlock_database("") == "❌ Please provide a database name"
    os.makedirs(os.path.join(r.db_root, "alpha"))
    os.makedirs(os.path.join(r.db_root, "checkpoints"))
    assert r.list_databases() == ["alpha"]
    assert r.load_database("alpha") == "❌ Collection not found in database: alpha"
    assert r.unlock_database("alpha").startswith("ℹ️ No lock file found")
    open(os.path.join(r.db_root, "alpha", ".lock"), "w").close()
    assert r.unlock_database("alpha") == "✅ Removed lock file from database: alpha"
    assert r.delete_database("alpha") == "✅ Deleted database: alpha"
    # guards of search_similar (core_system.py:652-653)
    assert r.search_similar()[0].startswith("❌ No query embeddings available")
    r.region_embeddings = [torch.zeros(4)]
    assert r.search_similar()[0].startswith("❌ No database loaded")
    r.request_stop()
    assert r._stop_requested


def test_manifest_of_delta_shards_parsing(tmp_path):
    """The append-only manifest of a collection (revers-o_amd/store.py): header, shard lines, the "complete" marker, rows
    appended after a save, and a torn last line (a crash while appending) that ends the parse without losing what stands."""
    man = str(tmp_path / "manifest.jsonl")
    lines = [{"format": 2, "collection": "c", "dim": 16},
             {"shard": 0, "file": "vectors.00000.f32.npy", "rows": 3, "ids": ["a", "b", "c"], "payloads": [{}, {}, {}], "files_done": ["x.jpg"]},
             {"shard": 1, "file": None, "rows": 0, "ids": [], "payloads": [], "files_done": ["broken.jpg"]}]
    with open(man, "w") as f:
        f.write("".join(json.dumps(x) + "\n" for x in lines))
    header, shards, complete, good = st.read_manifest(man)
    assert header["dim"] == 16 and [x["shard"] for x in shards] == [0, 1] and not complete and good == os.path.getsize(man)
    with open(man, "a") as f:
        f.write(json.dumps({"complete": True, "rows": 3}) + "\n")
    assert st.read_manifest(man)[2] is True
    whole = os.path.getsize(man)
    with open(man, "a") as f:
        f.write('{"shard": 2, "file": "vectors.0')                      # torn: no newline, not JSON
    header, shards, complete, good = st.read_manifest(man)
    assert complete and len(shards) == 2 and good == whole
    with open(man, "r+b") as f:
        f.truncate(good)
    with open(man, "a") as f:                                          # rows appended after a save: not complete until the next save
        f.write(json.dumps({"shard": 2, "file": "vectors.00002.f32.npy", "rows": 1, "ids": ["d"], "payloads": [{}], "files_done": []}) + "\n")
    assert st.read_manifest(man)[2] is False and len(st.read_manifest(man)[1]) == 3
    open(str(tmp_path / "empty.jsonl"), "w").close()
    with pytest.raises(ValueError):
        st.read_manifest(str(tmp_path / "empty.jsonl"))
    base = str(tmp_path / "checkpoints" / "db_checkpoint")
    os.makedirs(os.path.dirname(base))
    open(base + ".json", "w").write("{}")
    st.remove_checkpoint(base)
    assert not os.path.exists(base + ".json")


def test_region_metadata_rules(tmp_path):
    from reverso_amd.core_system import Regions
    from PIL import Image
    r = _bare_facade(tmp_path)
    pil = Image.new("RGB", (40, 30))
    mask = np.zeros((3, 30, 40), dtype=bool)
    mask[0, 5:10, 8:20] = True            # ordinary mask
    # mask[1] stays empty -> skipped (core_system.py:402-404)
    mask[2, 0:30, 0:40] = True
    reg = Regions([[8, 5, 19, 9], [0, 0, 1, 1], [0, 0, 39, 29]], mask=mask, confidence=[0.9, 0.8, 0.7],
                  class_id=[0, 1, 5], class_names=["person", "car"])
    kept, metas = r._region_metadata(pil, reg)
    assert kept == [0, 2]
    assert metas[0]["bbox"] == [8, 5, 19, 9] and metas[0]["detected_class"] == "person"
    assert abs(metas[0]["area_ratio"] - 60 / 1200) < 1e-9 and metas[0]["mask_status"] == "processed"
    assert metas[1]["detected_class"] == "object" and metas[1]["area_ratio"] == 1.0
    kept, metas = r._region_metadata(pil, Regions([[0, 0, 40, 30]]))      # no masks: full-image fallback
    assert metas[0]["bbox"] == [0, 0, 40, 30] and metas[0]["mask_status"] == "missing_or_unavailable"
    many = Regions(np.zeros((60, 4)))
    assert len(r._region_metadata(pil, many)[0]) == 50                      # cap (core_system.py:363)


def test_preprocess_matches_reference_transform():
    from PIL import Image
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    u8 = pp.resize_u8(arr, 28)
    assert u8.shape == (3, 28, 28) and u8.dtype == torch.uint8
    ref = np.asarray(Image.fromarray(arr).convert("RGB").resize((28, 28), Image.BILINEAR)).transpose(2, 0, 1)
    assert np.array_equal(u8.numpy(), ref)
    x = pp.normalize_u8(u8)
    assert torch.allclose(x, (torch.from_numpy(ref).float() / 255 - 0.5) / 0.5)
    assert float(x.min()) >= -1 and float(x.max()) <= 1
    same = pp.resize_u8(Image.fromarray(arr[:28, :28]), 28)                 # already at size: untouched
    assert np.array_equal(same.numpy(), arr[:28, :28].transpose(2, 0, 1))
    assert pp.batch_u8([arr, arr], 28).shape == (2, 3, 28, 28)


def test_preprocess_accepts_what_the_reference_accepts(tmp_path):
    from PIL import Image
    """core_system.py:435-439: an ndarray goes through Image.fromarray, a path through Image.open, anything else is taken
    as a PIL image, and every one is converted to RGB before the 336-px squash: grey, RGBA, palette and CMYK inputs end
    up as the same uint8 tensor the reference's transform would see."""
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (37, 53, 1), dtype=np.uint8)], -1)
    pal = Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=16)
    pal_path = os.path.join(str(tmp_path), "pal.png")
    pal.save(pal_path)
    cmyk_path = os.path.join(str(tmp_path), "cmyk.jpg")
    Image.fromarray(rgb).convert("CMYK").save(cmyk_path, quality=95)

    def want(pil):
        return np.asarray(pil.convert("RGB").resize((28, 28), Image.BILINEAR)).transpose(2, 0, 1)

    cases = [(grey, Image.fromarray(grey)), (rgba, Image.fromarray(rgba)), (pal_path, Image.open(pal_path)),
             (cmyk_path, Image.open(cmyk_path)), (Image.fromarray(rgb), Image.fromarray(rgb)), (pal, pal)]
    for given, ref in cases:
        got = pp.resize_u8(given, 28)
        assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 28, 28)
        assert np.array_equal(got.numpy(), want(ref))
    assert pp.to_pil(grey).mode == "RGB" and pp.to_pil(pal_path).size == (53, 37)


# --------------------------------------------------------------- gloo, world 2 ---
def _shard_worker(rank, world, port, tmp, N, D, Q, k, noise=0.0):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from reverso_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    gal = osearch.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    gal[10:14] = gal[10]                                # a tie group that straddles nothing but must stay ordered
    qs = osearch.normalize_rows(rng.standard_normal((Q, D), dtype=np.float32))
    sizes = [N // world + (1 if r < N % world else 0) for r in range(world)]
    lo = sum(sizes[:rank])
    mine = gal[lo:lo + sizes[rank]]

    backend = osearch.OracleShardBackend(mine, scan_noise=noise, row_offset=lo)
    ss = sharded.ShardedSearch(backend, sizes[rank])
    assert ss.offset == lo and ss.total_rows == N
    # data-parallel queries: each rank contributes its slice, gather gives everyone all of them
    per = Q // world
    allq = ss.gather_queries(torch.from_numpy(qs[rank * per:(rank + 1) * per]))
    assert np.array_equal(allq.numpy(), qs[: per * world])
    out = {}
    second_rounds = 0
    for thr in (None, 0.05):
        s, i, c = ss.search(allq, k, thr)
        out[str(thr)] = (s.numpy(), i.numpy(), c.numpy())
        second_rounds += ss.last_uncertified
    # a noisy scan (the bf16 scan's stand-in) must have sent some queries, not all, through the second, exact round;
    # an exact scan sends only queries whose k-th place ties with the candidate list's last (none here)
    assert (second_rounds > 0 and backend.exact_calls > 0) if noise > 0 else (second_rounds == 0 and backend.exact_calls == 0)
    # pipelined form: the next search is started before the previous one's result is asked for (no host wait between
    # searches).  A pending search whose shard state has been overwritten meanwhile re-does itself in step on every
    # rank if it needs the second round; same answers either way.
    ss.enable_timing(True)                                       # bench.py's `allgather_ms` / `exchanges_per_search`
    redone0, second0 = ss.redone_searches, ss.second_rounds
    pend = [ss.search_async(allq, k, thr) for thr in (None, 0.05)]
    for thr, p in zip((None, 0.05), pend):
        s, i, c = p.result()
        assert np.array_equal(i.numpy(), out[str(thr)][1]) and np.array_equal(c.numpy(), out[str(thr)][2])
        assert np.array_equal(s.numpy(), out[str(thr)][0])
        assert p.result()[1] is i                                # a finished search hands back the same tensors
    rep = ss.timing_report()
    ss.enable_timing(False)
    assert ss.timing_report() is None
    # this backend does not estimate: bounds + packed per search; a stale pending search (noisy scan: both are -- the
    # second search overwrote the first one's candidates, the first one's repair the second one's) searches only its
    # uncertified queries again, a small search of its own with ITS second round: never the whole batch, no cascade
    assert rep["allgather_ms"]["bounds"] > 0 and rep["allgather_ms"]["packed"] > 0
    if noise > 0:
        redone = ss.redone_searches - redone0
        assert redone == 2 and ss.second_rounds - second0 == 2, (ss.redone_searches, ss.second_rounds)
        assert rep["searches"] == 2 + redone and "packed_second_round" in rep["allgather_ms"]
        assert 2.0 < rep["exchanges_per_search"] <= 3.0
    else:
        assert ss.redone_searches == redone0 and rep["searches"] == 2 and rep["exchanges_per_search"] == 2.0
    # Eleven searches pending, none asked for (more than the eight landing places): the ninth has to finish the first before
    # it takes over its slot -- and the first, uncertified under a noisy scan, runs a repair search of its own inside that
    # flush.  Each search asks a different subset of the queries, so a result filed under the wrong search (or second rounds
    # run against another search's candidates) shows as wrong rows.
    subsets = [list(range(j % 3, per * world, 1 + j % 2)) for j in range(11)]
    pend = [ss.search_async(allq[sub].contiguous(), k, None) for sub in subsets]
    for sub, p in reversed(list(zip(subsets, pend))):
        s, i, c = p.result()
        assert np.array_equal(i.numpy(), out["None"][1][sub]) and np.array_equal(c.numpy(), out["None"][2][sub])
        assert np.array_equal(s.numpy(), out["None"][0][sub])
    assert all(x is None for x in ss._inflight)
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **{f"{t}_{n}": v for t, (a, b, c) in out.items()
                                                        for n, v in (("s", a), ("i", b), ("c", c))})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("noise", [0.0, 0.1])
def test_sharded_search_protocol_gloo(tmp_path, noise):
    """World-size-2 run of the sharded search's protocol (offsets, both all-gathers, the merge, the certificate check
    and -- with a noisy scan -- the second, exact round) on CPU: the result must be the exhaustive search's."""
    import torch.multiprocessing as mp
    N, D, Q, k, world = 1001, 64, 6, 7, 2
    port = 29500 + (os.getpid() % 2000) + (7 if noise else 0)
    mp.spawn(_shard_worker, args=(world, port, str(tmp_path), N, D, Q, k, noise), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    gal = osearch.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    gal[10:14] = gal[10]
    qs = osearch.normalize_rows(rng.standard_normal((Q, D), dtype=np.float32))
    for thr in (None, 0.05):
        rs, ri, rc = osearch.search(gal, qs, k, thr, normalize=False)
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
            assert np.array_equal(got[f"{thr}_i"], ri)
            assert np.array_equal(got[f"{thr}_c"], rc)
            np.testing.assert_allclose(got[f"{thr}_s"], rs, atol=1e-6)


@pytest.mark.parametrize("noise,shards", [(0.0, 3), (0.25, 4)])
def test_local_shards_protocol_on_cpu(noise, shards):
    """sharded.LocalShards -- the two-phase protocol with its certificate round over several shards held by one process
    (one handle per GPU, or the GPU tests' eight shards on one device) -- with the numpy backend: equals the exhaustive
    search, with and without a noisy scan stand-in, thresholds included, uneven shard sizes, k larger than a shard."""
    from reverso_amd import sharded
    rng = np.random.default_rng(21)
    N, D, Q, k = 700, 32, 9, 12
    gal = osearch.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    gal[40:48] = gal[40]
    qs = osearch.normalize_rows(rng.standard_normal((Q, D), dtype=np.float32))
    qs[0] = gal[40]
    cuts = [0] + sorted(rng.choice(np.arange(20, N - 20), size=shards - 1, replace=False).tolist()) + [N]
    backends = [osearch.OracleShardBackend(gal[a:b], scan_noise=noise, row_offset=a) for a, b in zip(cuts[:-1], cuts[1:])]
    ls = sharded.LocalShards(backends, cuts[:-1])
    second = 0
    for thr in (None, 0.2):
        s, i, c = ls.search(torch.from_numpy(qs), k, thr)
        second += ls.last_uncertified
        rs, ri, rc = osearch.search(gal, qs, k, thr, normalize=False)
        assert np.array_equal(i.numpy(), ri) and np.array_equal(c.numpy(), rc)
        np.testing.assert_allclose(s.numpy(), rs, atol=1e-6)
    assert (second > 0) == (noise > 0)


def test_swap_in_and_recover_after_a_crash_in_the_final_swap(tmp_path):
    """The finished build replaces the database by two renames (old -> <db>.revo-old, build -> <db>); whatever a crash between
    them leaves behind, recover() -- run by list_databases / load_database -- puts a COMPLETE directory back under the
    database's name, and never adopts an unfinished build."""
    import shutil

    def make(path, complete, tag):
        os.makedirs(path)
        with open(os.path.join(path, st.MANIFEST), "w") as f:
            f.write(json.dumps({"format": 2, "collection": tag, "dim": 64}) + "\n")
            if complete:
                f.write(json.dumps({"complete": True, "rows": 0}) + "\n")
    db = str(tmp_path / "db")
    # the normal swap, with and without an old database
    make(db + ".building", True, "new1")
    st.swap_in(db + ".building", db)
    assert st.read_manifest(os.path.join(db, st.MANIFEST))[0]["collection"] == "new1" and not os.path.exists(db + ".building")
    make(db + ".building", True, "new2")
    st.swap_in(db + ".building", db)
    assert st.read_manifest(os.path.join(db, st.MANIFEST))[0]["collection"] == "new2"
    assert not os.path.exists(db + st.OLD) and not os.path.exists(db + ".building")
    # a build that another process's recover() adopted between the last manifest line and the swap: nothing left to do
    make(db + ".building", True, "adopted")
    shutil.rmtree(db)
    assert st.recover(db) == "build"
    st.swap_in(db + ".building", db)                              # (used to fail with FileNotFoundError)
    assert st.read_manifest(os.path.join(db, st.MANIFEST))[0]["collection"] == "adopted"
    make(db + ".building", True, "new2")
    st.swap_in(db + ".building", db)
    # crash after "old -> .revo-old", before "build -> db": the complete build is adopted
    make(db + ".building", True, "new3")
    os.replace(db, db + st.OLD)
    assert st.recover(db) == "build"
    assert st.read_manifest(os.path.join(db, st.MANIFEST))[0]["collection"] == "new3"
    assert st.recover(db) is None                                # nothing to do when the database is there
    # the database is gone and only an UNFINISHED build and the set-aside old one exist: the old one comes back
    shutil.rmtree(db)
    make(db + ".building", False, "half")
    assert st.recover(db) == "old"
    assert st.read_manifest(os.path.join(db, st.MANIFEST))[0]["collection"] == "new2" and os.path.isdir(db + ".building")
    # only an unfinished build: nothing is adopted
    shutil.rmtree(db)
    assert st.recover(db) is None and not os.path.exists(db)


def test_legacy_set_aside_suffix_is_recovered_hidden_and_deleted(tmp_path):
    """A crash under an earlier build left <db>.old (the suffix was renamed to .revo-old): for one release recover() still
    adopts it, list_databases hides it and delete_database removes it -- but a user's own directory called "x.old" that
    is not one of this package's databases stays a database."""
    r = _bare_facade(tmp_path)
    os.makedirs(r.db_root)

    def make(name, complete=True):
        path = os.path.join(r.db_root, name)
        os.makedirs(path)
        with open(os.path.join(path, st.MANIFEST), "w") as f:
            f.write(json.dumps({"format": 2, "collection": name, "dim": 64}) + "\n")
            if complete:
                f.write(json.dumps({"complete": True, "rows": 0}) + "\n")
    make("alpha.old")                                    # the crash: alpha itself is gone
    os.makedirs(os.path.join(r.db_root, "notes.old"))    # a user's directory (no manifest): never touched
    assert sorted(r.list_databases()) == ["alpha", "notes.old"]
    assert not os.path.exists(os.path.join(r.db_root, "alpha.old"))
    make("alpha.old")                                    # a swap that died before the set-aside copy was removed
    assert sorted(r.list_databases()) == ["alpha", "notes.old"]
    assert r.delete_database("alpha").startswith("✅")
    assert not os.path.exists(os.path.join(r.db_root, "alpha.old")) and os.path.isdir(os.path.join(r.db_root, "notes.old"))


def test_overlapping_builds_keep_their_names_until_each_returns(tmp_path):
    """Two create_database calls on one instance (UI threads): the first to finish must not clear the other's entry."""
    r = _bare_facade(tmp_path)
    r._building_names, r._building_lock = set(), __import__("threading").Lock()
    import threading
    gate_a, gate_b = threading.Event(), threading.Event()
    seen = {}

    def fake(self_, folder, name, *a):
        if name == "a":
            gate_a.wait(5)
        else:
            seen["a_while_b"] = r._is_building("a")
            gate_b.wait(5)
        return name
    r._create_database = fake.__get__(r)
    ta = threading.Thread(target=lambda: r.create_database("f", "a"))
    tb = threading.Thread(target=lambda: r.create_database("f", "b"))
    ta.start(); tb.start()
    import time
    for _ in range(200):
        if r._is_building("a") and r._is_building("b"):
            break
        time.sleep(0.01)
    assert r._is_building("a") and r._is_building("b")
    gate_a.set(); ta.join(5)
    assert not r._is_building("a") and r._is_building("b")        # (was: one slot, reset by the first finisher)
    gate_b.set(); tb.join(5)
    assert not r._is_building("b") and seen["a_while_b"]
