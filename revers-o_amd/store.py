"""Gallery store: device-resident vectors + host-side ids/payloads + persistence.

Stands where the reference keeps a ``QdrantClient(path=...)`` with one COSINE
collection (``core_system.py:100``, ``:521``, ``:600-603``, ``:608-622``, ``:659-664``).
Qdrant's sqlite format is not a compatibility target (third-party, un-pinned;
SURVEY.md §8(f) row 2): a database directory here holds

    vectors.f32.npy     normalised fp32 rows (the gallery's master copy)
    meta.json           {"collection", "dim", "ids": [...], "payloads": [...]}
    .lock               present while a process has the database open

and ``checkpoints/<name>_checkpoint.{json,npy}`` make the reference's (inoperative,
``core_system.py:480-489``, ``:524-538``) resume actually work.
"""
import json
import os
from dataclasses import dataclass

import numpy as np
import torch

from .engine import Gallery


@dataclass
class ScoredPoint:
    """Shape of a qdrant search hit as the reference consumes it (core_system.py:671-676)."""
    id: str
    score: float
    payload: dict


class GalleryStore:
    def __init__(self, dim, device=0, capacity=65536, collection="simple_reverso", path=None):
        self.dim = int(dim)
        self.device = device
        self.collection = collection
        self.path = path
        self.ids = []
        self.payloads = []
        self.gallery = Gallery(self.dim, max(int(capacity), 1), device=device)
        if path:
            os.makedirs(path, exist_ok=True)
            open(os.path.join(path, ".lock"), "a").close()

    def __len__(self):
        return len(self.ids)

    def _grow(self, need):
        if need <= self.gallery.capacity:
            return
        cap = max(need, 2 * self.gallery.capacity)
        old = self.gallery
        new = Gallery(self.dim, cap, device=self.device)
        n = len(old)
        if n:
            new.add(old.read(0, n), normalize=False)
        old.close()
        self.gallery = new

    def upsert(self, vectors, ids, payloads):
        """vectors: [n, dim] fp32 tensor (host or device); rows are normalised at insert."""
        vectors = torch.as_tensor(vectors, dtype=torch.float32)
        assert vectors.shape[0] == len(ids) == len(payloads)
        self._grow(len(self) + vectors.shape[0])
        self.gallery.add(vectors, normalize=True)
        self.ids.extend(ids)
        self.payloads.extend(payloads)

    def search(self, query_vector, limit, score_threshold=None):
        """One query, qdrant-style result list (core_system.py:659-664)."""
        q = torch.as_tensor(query_vector, dtype=torch.float32).reshape(1, -1)
        dev = self.gallery.device
        s, i, c = self.gallery.search(q.to(dev), k=int(limit), score_threshold=score_threshold)
        n = int(c[0])
        s, i = s[0, :n].tolist(), i[0, :n].tolist()
        return [ScoredPoint(self.ids[j], float(sc), self.payloads[j]) for sc, j in zip(s, i)]

    # -- persistence ----------------------------------------------------------
    def save(self, path=None):
        path = path or self.path
        os.makedirs(path, exist_ok=True)
        n = len(self)
        vec = self.gallery.read(0, n).cpu().numpy() if n else np.zeros((0, self.dim), np.float32)
        np.save(os.path.join(path, "vectors.f32.npy"), vec)
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump({"collection": self.collection, "dim": self.dim, "ids": self.ids, "payloads": self.payloads}, f)

    @classmethod
    def load(cls, path, device=0):
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        vec = np.load(os.path.join(path, "vectors.f32.npy"))
        st = cls(meta["dim"], device=device, capacity=max(len(meta["ids"]), 1), collection=meta["collection"], path=path)
        if len(meta["ids"]):
            st.gallery.add(torch.from_numpy(vec), normalize=False)     # stored rows are already normalised
        st.ids, st.payloads = list(meta["ids"]), list(meta["payloads"])
        return st

    def close(self):
        if self.path:
            try:
                os.remove(os.path.join(self.path, ".lock"))
            except OSError:
                pass
        self.gallery.close()


# -- checkpoint / resume of a gallery build (core_system.py:474-489, :524-538) ----
def save_checkpoint(ckpt_base, processed_files, embeddings, metadata, database_name, folder_path):
    """embeddings: list of fp32 CPU tensors [D]."""
    from datetime import datetime
    os.makedirs(os.path.dirname(ckpt_base), exist_ok=True)
    arr = torch.stack(embeddings).numpy() if embeddings else np.zeros((0, 0), np.float32)
    np.save(ckpt_base + ".npy.tmp.npy", arr)
    os.replace(ckpt_base + ".npy.tmp.npy", ckpt_base + ".npy")
    with open(ckpt_base + ".json.tmp", "w") as f:
        json.dump({"processed_files": sorted(processed_files), "timestamp": datetime.now().isoformat(),
                   "database_name": database_name, "folder_path": folder_path, "partial_metadata": metadata,
                   "n_embeddings": len(embeddings)}, f, indent=2)
    os.replace(ckpt_base + ".json.tmp", ckpt_base + ".json")


def load_checkpoint(ckpt_base):
    with open(ckpt_base + ".json") as f:
        data = json.load(f)
    arr = np.load(ckpt_base + ".npy")
    if arr.shape[0] != data["n_embeddings"] or arr.shape[0] != len(data["partial_metadata"]):
        raise ValueError("checkpoint vectors and metadata disagree")
    embs = [torch.from_numpy(arr[i].copy()) for i in range(arr.shape[0])]
    return set(data["processed_files"]), embs, data["partial_metadata"]


def remove_checkpoint(ckpt_base):
    for ext in (".json", ".npy"):
        try:
            os.remove(ckpt_base + ext)
        except OSError:
            pass
