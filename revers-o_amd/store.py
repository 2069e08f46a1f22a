"""Gallery store: device-resident vectors + host-side ids/payloads + persistence.

Stands where the reference keeps a ``QdrantClient(path=...)`` with one COSINE
collection (``core_system.py:100``, ``:521``, ``:600-603``, ``:608-622``, ``:659-664``).
Qdrant's sqlite format is not a compatibility target (third-party, un-pinned;
SURVEY.md §8(f) row 2).  A database directory here is append-only:

    manifest.jsonl          line 1: {"format": 2, "collection", "dim"}; then one line per DELTA SHARD, in order:
                            {"shard": i, "file", "rows", "ids": [...], "payloads": [...], "files_done": [...]};
                            a last line {"complete": true, "rows": N} once a build has finished
    vectors.00000.f32.npy   the shard's normalised fp32 rows (the gallery's master copy)
    .lock                   present while a process has the database open

A save / checkpoint writes only the rows added since the last one (one .npy + one manifest line, the .npy first and
under a temporary name: a crash leaves at worst an orphan file no line points to) -- never the whole set: building a
1 M-vector gallery writes 4 GB once, not 4 GB per checkpoint.  The manifest of an unfinished build (no "complete"
line) IS its checkpoint: ``files_done`` says which source files the shards' rows cover (``create_database`` resumes
from there; the reference's own checkpoint is inoperative, ``core_system.py:480-489``, ``:524-538``).
Round-1/2 directories (one ``vectors.f32.npy`` + ``meta.json``) still load.
"""
import json
import os
import shutil
from dataclasses import dataclass

import numpy as np
import torch

from .engine import Gallery

MANIFEST = "manifest.jsonl"


def _fsync_dir(path):
    """Make a rename / create inside `path` durable (no-op where directories cannot be opened)."""
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return
    try:
        os.fsync(fd)
    except OSError:
        pass
    finally:
        os.close(fd)


OLD = ".revo-old"          # the set-aside previous database during a swap (a suffix no user database name is likely to end in)
OLD_LEGACY = ".old"        # what earlier builds called it: a crash under one of those may have left <db>.old behind -- recover(),
                           # list_databases and delete_database recognise it for one release (only next to a database name that
                           # is missing or was made by this package: a user database that happens to be called "x.old" and has
                           # no sibling "x" story is left alone, see is_legacy_set_aside)


def is_legacy_set_aside(root, name):
    """True if directory ``name`` = "<db>.old" under ``root`` is a previous build's set-aside copy rather than a user's
    database: it carries this package's manifest and either <db> is missing (the crash it was left by) or <db> is one of
    this package's databases too (a swap that died before the set-aside copy was removed)."""
    if not name.endswith(OLD_LEGACY) or name.endswith(OLD):
        return False
    base = os.path.join(root, name[: -len(OLD_LEGACY)])
    if not os.path.isfile(os.path.join(root, name, MANIFEST)):
        return False
    return (not os.path.isdir(base)) or os.path.isfile(os.path.join(base, MANIFEST))


def _is_complete(path):
    man = os.path.join(path, MANIFEST)
    try:
        return os.path.isfile(man) and bool(read_manifest(man)[2])
    except (OSError, ValueError):
        return False


def swap_in(build_path, db_path):
    """Replace the database directory by the finished build without a moment in which neither exists under a name
    list_databases / load_database look at: old -> <db>.revo-old, build -> <db>, then the old one is removed.  A crash in
    between leaves <db>.revo-old (complete) and possibly <db>.building (complete): recover() puts things right.
    If another process's recover() has already adopted the (complete) build under the database's name -- it can, between
    this build's last manifest line and this call -- there is nothing left to do."""
    old = db_path + OLD
    if not os.path.isdir(build_path) and _is_complete(db_path):
        return
    if os.path.isdir(old):
        shutil.rmtree(old)
    had = os.path.isdir(db_path)
    if had:
        os.replace(db_path, old)
    os.replace(build_path, db_path)
    _fsync_dir(os.path.dirname(db_path) or ".")
    if had:
        shutil.rmtree(old, ignore_errors=True)


def recover(db_path, building_suffix=".building"):
    """After a crash inside swap_in: a database that is missing while a COMPLETE build (or the set-aside old one) sits
    next to it is put back under its name.  Returns what was adopted, or None."""
    if os.path.isdir(db_path):
        return None
    for cand, what in ((db_path + building_suffix, "build"), (db_path + OLD, "old"), (db_path + OLD_LEGACY, "old")):
        man = os.path.join(cand, MANIFEST)
        if os.path.isfile(man):
            try:
                complete = read_manifest(man)[2]
            except (OSError, ValueError):
                complete = False
            if complete:
                os.replace(cand, db_path)
                return what
    return None


def read_manifest(man):
    """Parse a manifest.  Returns (header, shard records in order, complete, bytes of the file that are whole lines).
    A torn last line -- the process died while appending -- ends the parse: everything before it stands."""
    header, shards, complete, good = None, [], False, 0
    with open(man, "rb") as f:
        for raw in f:
            if not raw.endswith(b"\n"):
                break
            ln = raw.strip()
            if ln:
                try:
                    rec = json.loads(ln)
                except ValueError:
                    break
                if header is None:
                    header = rec
                elif rec.get("complete"):
                    complete = True
                else:
                    complete = False                   # rows appended after a save: complete again at the next save
                    shards.append(rec)
            good += len(raw)
    if header is None:
        raise ValueError(f"{man}: empty manifest")
    return header, shards, complete, good


@dataclass
class ScoredPoint:
    """Shape of a qdrant search hit as the reference consumes it (core_system.py:671-676)."""
    id: str
    score: float
    payload: dict


class GalleryStore:
    def __init__(self, dim, device=0, capacity=65536, collection="simple_reverso", path=None, _fresh=True, build_info=None):
        """``build_info``: what the vectors were made from and how (source folder, model, region mode, ...), written into
        the manifest header; a resume compares it (core_system.create_database) so that rows of different builds never mix."""
        self.build_info = dict(build_info or {})
        self.dim = int(dim)
        self.device = device
        self.collection = collection
        self.path = path
        self.ids = []
        self.payloads = []
        self.gallery = Gallery(self.dim, max(int(capacity), 1), device=device)
        self.complete = False
        self._flushed = 0            # rows already in shards on disk
        self._shards = 0
        self._files_pending = []     # source files whose rows were added since the last flush
        self.files_done = set()      # source files covered by the shards on disk
        if path:
            os.makedirs(path, exist_ok=True)
            open(os.path.join(path, ".lock"), "a").close()
            if _fresh:
                with open(os.path.join(path, MANIFEST), "w") as f:
                    f.write(json.dumps({"format": 2, "collection": collection, "dim": self.dim, "build": self.build_info}) + "\n")

    def __len__(self):
        return len(self.ids)

    def _grow(self, need):
        if need <= self.gallery.capacity:
            return
        cap = max(need, 2 * self.gallery.capacity)         # geometric: a build of N rows copies < 2 N rows in all
        old = self.gallery
        new = Gallery(self.dim, cap, device=self.device)
        n = len(old)
        step = max(1, (128 << 20) // (self.dim * 4))       # 128 MB of fp32 rows at a time: the peak is old + new + one chunk,
        for s0 in range(0, n, step):                       # not old + new + a full fp32 copy of old
            new.add(old.read(s0, min(step, n - s0)), normalize=False)       # device to device
        old.close()
        self.gallery = new

    def upsert(self, vectors, ids, payloads, files=None):
        """vectors: [n, dim] fp32 tensor, host or DEVICE (an ingest appends its embeddings where they are: they never
        visit the host); rows are normalised at insert.  ``files``: source files these rows complete (resume bookkeeping)."""
        vectors = torch.as_tensor(vectors, dtype=torch.float32)
        assert vectors.shape[0] == len(ids) == len(payloads)
        if vectors.shape[0]:
            self._grow(len(self) + vectors.shape[0])
            self.gallery.add(vectors, normalize=True)
            self.ids.extend(ids)
            self.payloads.extend(payloads)
        if files:
            self._files_pending.extend(files)
        self.complete = False

    def search(self, query_vector, limit, score_threshold=None):
        """One query, qdrant-style result list (core_system.py:659-664)."""
        q = torch.as_tensor(query_vector, dtype=torch.float32).reshape(1, -1)
        dev = self.gallery.device
        s, i, c = self.gallery.search(q.to(dev), k=int(limit), score_threshold=score_threshold)
        n = int(c[0])
        s, i = s[0, :n].tolist(), i[0, :n].tolist()
        return [ScoredPoint(self.ids[j], float(sc), self.payloads[j]) for sc, j in zip(s, i)]

    # -- persistence ----------------------------------------------------------
    def flush(self, path=None):
        """Write the rows (and finished source files) added since the last flush as one delta shard.  Returns the rows written."""
        path = path or self.path
        if not path:
            raise ValueError("this store has no directory to flush to")
        n = len(self)
        new = n - self._flushed
        if new <= 0 and not self._files_pending:
            return 0
        name = None
        if new > 0:
            name = f"vectors.{self._shards:05d}.f32.npy"
            vec = self.gallery.read(self._flushed, new).cpu().numpy()
            tmp = os.path.join(path, name + ".tmp.npy")
            with open(tmp, "wb") as f:                     # the shard's bytes and its name are on disk BEFORE the manifest
                np.save(f, vec)                            # line that points to them is appended: after a power loss the
                f.flush()                                  # manifest never names a shard that is not there
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(path, name))
            _fsync_dir(path)
        line = {"shard": self._shards, "file": name, "rows": new, "ids": self.ids[self._flushed:n],
                "payloads": self.payloads[self._flushed:n], "files_done": self._files_pending}
        with open(os.path.join(path, MANIFEST), "a") as f:
            f.write(json.dumps(line) + "\n")
            f.flush()
            os.fsync(f.fileno())
        self.files_done.update(self._files_pending)
        self._files_pending = []
        self._flushed = n
        self._shards += 1
        return new

    def save(self, path=None):
        """Flush what is new and mark the collection complete."""
        path = path or self.path
        if not path:
            raise ValueError("this store has no directory (created without one, or loaded from the one-file format of "
                             "rounds 1-2): save(path=<new directory>) writes it out in the delta-shard format")
        os.makedirs(path, exist_ok=True)
        if path != self.path:                              # saving somewhere else: write everything there
            other = os.path.join(path, MANIFEST)
            with open(other, "w") as f:
                f.write(json.dumps({"format": 2, "collection": self.collection, "dim": self.dim, "build": self.build_info}) + "\n")
            keep = (self._flushed, self._shards, self._files_pending, self.path)
            self._flushed, self._shards, self._files_pending = 0, 0, sorted(self.files_done) + self._files_pending
            try:
                self.flush(path)
                with open(other, "a") as f:
                    f.write(json.dumps({"complete": True, "rows": len(self)}) + "\n")
            finally:
                self._flushed, self._shards, self._files_pending, self.path = keep
            return
        self.flush(path)
        if not self.complete:
            with open(os.path.join(path, MANIFEST), "a") as f:
                f.write(json.dumps({"complete": True, "rows": len(self)}) + "\n")
                f.flush()
                os.fsync(f.fileno())
            self.complete = True

    @classmethod
    def load(cls, path, device=0, allow_partial=False, capacity=0):
        """Open a database directory.  An unfinished build (no "complete" line) raises unless ``allow_partial``
        (``create_database(resume_from_checkpoint=True)`` continues it)."""
        man = os.path.join(path, MANIFEST)
        if not os.path.exists(man):
            return cls._load_v1(path, device)
        header, shards, complete, good_bytes = read_manifest(man)
        if not complete and not allow_partial:
            raise ValueError(f"{path}: unfinished build (resume it with create_database(resume_from_checkpoint=True))")
        if good_bytes < os.path.getsize(man):
            # a torn last line (the process died while appending): cut it off before anything is appended behind it
            with open(man, "r+b") as f:
                f.truncate(good_bytes)
        rows = sum(s["rows"] for s in shards)
        st = cls(header["dim"], device=device, capacity=max(rows, capacity, 1), collection=header["collection"], path=path,
                 _fresh=False, build_info=header.get("build"))
        for s in shards:
            if s["rows"]:
                vec = np.load(os.path.join(path, s["file"]))
                if vec.shape != (s["rows"], st.dim):
                    raise ValueError(f"{path}/{s['file']}: {vec.shape} does not match its manifest line")
                st.gallery.add(torch.from_numpy(vec), normalize=False)      # stored rows are already normalised
                st.ids.extend(s["ids"])
                st.payloads.extend(s["payloads"])
            st.files_done.update(s.get("files_done", []))
        st._flushed, st._shards, st.complete = rows, len(shards), complete
        return st

    @classmethod
    def _load_v1(cls, path, device):
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        vec = np.load(os.path.join(path, "vectors.f32.npy"))
        st = cls(meta["dim"], device=device, capacity=max(len(meta["ids"]), 1), collection=meta["collection"], path=None)
        if len(meta["ids"]):
            st.gallery.add(torch.from_numpy(vec), normalize=False)
        st.ids, st.payloads = list(meta["ids"]), list(meta["payloads"])
        st.complete = True
        return st

    def close(self):
        if self.path:
            try:
                os.remove(os.path.join(self.path, ".lock"))
            except OSError:
                pass
        self.gallery.close()


# -- the small JSON at the reference's checkpoint path (core_system.py:474-476); the vectors live in the build's shards
def remove_checkpoint(ckpt_base):
    for ext in (".json", ".npy"):
        try:
            os.remove(ckpt_base + ext)
        except OSError:
            pass
