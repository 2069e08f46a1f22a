"""``SimpleReverso`` façade over the MI355X-native embed + search path.

Same class name, method names, argument meaning, return shapes and ``❌ / ⚠️``
string conventions as the reference's ``core_system.py`` (class at ``:44``), so the
reference's ``ui.py`` / ``main.py`` run on top of it unchanged.
What differs underneath:

* ``encode_image`` (``:341``, ``:442``) is the HIP forward in ``librevo.so``;
* the vector store (``QdrantClient``, ``:100``, ``:521``, ``:600-622``, ``:659-664``) is a
  device-resident gallery with bit-exact fp32 re-scored top-k;
* ``create_database`` embeds in batches (decode threads overlap the device) instead
  of one image per forward (``:541-591``), and its checkpoint/resume works;
* GroundedSAM (``:205-318``) is third-party and out of scope: a detector callable can be
  injected (``detector=``); without one every image is one full-frame region, which
  is also what the reference's embedding of a region amounts to (``:406`` "Use global
  for now": every region receives the global embedding).

There is no CPU fallback: constructing the class without a GPU raises.
"""
import json
import os
from datetime import datetime
import shutil
import threading
import time
import random
import uuid
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont

from . import preprocess as pp
from . import store as st
from .config import DEFAULT_VARIANT, available_configs, get_config
from .engine import VitEngine
from .weights import load_state_dict, synth_weights

DB_ROOT = "./simple_reverso_db"
IMAGE_EXTENSIONS = ['.jpg', '.jpeg', '.png', '.bmp', '.tiff', '.webp']
BUILDING = ".building"      # suffix of the directory a collection is built in (renamed over the old one when complete)



_uuid_rng = random.Random()   # ids only have to be unique (uuid4 format as upstream), not unpredictable


def _uuid4():
    """uuid.uuid4() without the os.urandom system call (35 us each, three per stored region)"""
    return str(uuid.UUID(int=_uuid_rng.getrandbits(128), version=4))

class Regions:
    """Minimal stand-in for ``supervision.Detections`` (xyxy, mask, confidence, class_id)."""

    def __init__(self, xyxy, mask=None, confidence=None, class_id=None, class_names=None):
        self.xyxy = np.asarray(xyxy, dtype=np.float32).reshape(-1, 4)
        self.mask = mask
        n = len(self.xyxy)
        self.confidence = np.ones(n, np.float32) if confidence is None else np.asarray(confidence, np.float32)
        self.class_id = np.zeros(n, np.int64) if class_id is None else np.asarray(class_id, np.int64)
        self.class_names = class_names or ["object"]

    def __len__(self):
        return len(self.xyxy)


class SimpleReverso:
    """Simplified visual investigation system (MI355X-native hot path)."""
    # names of the databases create_database calls of this instance are building right now (two UI threads may overlap):
    # list_databases / load_database leave those build directories alone meanwhile.  Guarded by _building_lock.
    _building_lock = threading.Lock()
    _building_names = frozenset()      # (per-instance set from __init__ on)

    def __init__(self, model_name=DEFAULT_VARIANT, checkpoint=None, device=0, db_root=DB_ROOT, max_batch=64,
                 detector=None, decode_workers=None, synthetic_seed=0, region_mode="global", device_resize=False,
                 checkpoint_interval_s=30.0):
        print("🚀 Initializing Simple Revers-o...")
        if region_mode not in ("global", "crop"):
            raise ValueError("region_mode must be 'global' (the reference's behaviour, core_system.py:406) or 'crop'")
        # "crop": every region is cropped to its box on the device and embedded on its own -- the
        # feature the reference leaves as a placeholder (:406 "Use global for now", :687-690)
        self.region_mode = region_mode
        # True: decoded frames are uploaded as they are and squash-resized by the HIP kernel
        # (same pixels as the host PIL resize, bit for bit); False: PIL resize in the decode pool
        self.device_resize = bool(device_resize)
        self.checkpoint_interval_s = float(checkpoint_interval_s)     # a checkpoint writes the vectors added since the last one (a delta shard)
        self.db_root = db_root
        self.max_batch = int(max_batch)
        self.detector = detector
        # decode threads of a gallery build.  None = by mode (create_database): MORE threads are not better -- PIL's
        # RGBX -> RGB packing of a decoded frame runs under the GIL (0.35 ms per 640 x 480 frame), and the ingest thread,
        # which has to keep the device's queue full, waits its turn behind every thread that wants it: measured on
        # 3 000 JPEGs with device resize 1 950 images/s with 8 threads, 1 620 with 16, 1 720 with 32.  (Decode worker
        # PROCESSES writing into pinned shared memory were built and measured too: 1 975 and 563-598 images/s against
        # 1 950 and 602-649 with threads -- with the right thread count the build waits for the device, not for the GIL.)
        self.decode_workers = None if decode_workers is None else int(decode_workers)
        self._decode_pool = ThreadPoolExecutor(max_workers=self.decode_workers or 8)
        self._lock = threading.RLock()      # ui.py drives one shared instance from worker threads
        self.device = self.setup_device(device)
        self.pe_model, self.preprocess = self.load_pe_model(model_name, checkpoint, synthetic_seed)
        self.grounded_sam = None
        self.vector_db = None
        self.current_database = None
        self.detected_regions = []
        self.region_embeddings = None
        self.query_embedding_for_search = None
        self._building_names = set()
        self._building_lock = threading.Lock()
        self._stop_requested = False
        self._last_processed_file = None
        self._partial_embeddings = []
        self._partial_metadata = []
        self._build_store = None        # the collection a create_database call is building (closed when the call returns)
        self.last_ingest_stats = None   # wall-clock stage times of the last create_database call
        print("✅ Simple Revers-o ready!")

    # ------------------------------------------------------------- DB admin --
    def list_databases(self):
        """core_system.py:74-88"""
        if not os.path.exists(self.db_root):
            return []
        for n in os.listdir(self.db_root):              # a crash inside the final swap of a build: put the database back
            for suf in (BUILDING, st.OLD, st.OLD_LEGACY):
                if suf == st.OLD_LEGACY and not st.is_legacy_set_aside(self.db_root, n):
                    continue
                # (not a build this very process is finishing: create_database swaps that one in itself)
                if n.endswith(suf) and not self._is_building(n[: -len(suf)]):
                    st.recover(os.path.join(self.db_root, n[: -len(suf)]), BUILDING)
        return [n for n in os.listdir(self.db_root)
                if os.path.isdir(os.path.join(self.db_root, n)) and n != "checkpoints" and not n.endswith(BUILDING)
                and not n.endswith(st.OLD) and not st.is_legacy_set_aside(self.db_root, n)]

    def _is_building(self, name):
        with self._building_lock:
            return name in self._building_names

    def load_database(self, database_name):
        """core_system.py:90-119"""
        if not database_name:
            return "❌ Please provide a database name"
        db_path = os.path.join(self.db_root, database_name)
        if not self._is_building(database_name):
            st.recover(db_path, BUILDING)
        if not os.path.exists(db_path):
            return f"❌ Database not found: {database_name}"
        try:
            if not (os.path.exists(os.path.join(db_path, st.MANIFEST)) or os.path.exists(os.path.join(db_path, "meta.json"))):
                return f"❌ Collection not found in database: {database_name}"
            with self._lock:
                if self.vector_db is not None:
                    self.vector_db.close()
                self.vector_db = st.GalleryStore.load(db_path, device=self.device.index or 0)
                self.current_database = self.vector_db.collection
            return f"✅ Loaded database: {database_name}"
        except Exception as e:
            return f"❌ Error loading database: {str(e)}"

    def delete_database(self, database_name):
        """core_system.py:121-135"""
        if not database_name:
            return "❌ Please provide a database name"
        db_path = os.path.join(self.db_root, database_name)
        if not os.path.exists(db_path):
            return f"❌ Database not found: {database_name}"
        try:
            shutil.rmtree(db_path)
            # an unfinished build of the same name and its checkpoint note go with it
            shutil.rmtree(db_path + BUILDING, ignore_errors=True)
            shutil.rmtree(db_path + st.OLD, ignore_errors=True)
            if st.is_legacy_set_aside(self.db_root, database_name + st.OLD_LEGACY):
                shutil.rmtree(db_path + st.OLD_LEGACY, ignore_errors=True)
            st.remove_checkpoint(os.path.join(self.db_root, "checkpoints", f"{database_name}_checkpoint"))
            return f"✅ Deleted database: {database_name}"
        except Exception as e:
            return f"❌ Error deleting database: {str(e)}"

    def unlock_database(self, database_name):
        """core_system.py:137-154"""
        if not database_name:
            return "❌ Please provide a database name"
        db_path = os.path.join(self.db_root, database_name)
        if not os.path.exists(db_path):
            return f"❌ Database not found: {database_name}"
        lock_file = os.path.join(db_path, ".lock")
        if os.path.exists(lock_file):
            try:
                os.remove(lock_file)
                return f"✅ Removed lock file from database: {database_name}"
            except Exception as e:
                return f"❌ Error removing lock file: {str(e)}"
        return f"ℹ️ No lock file found for database: {database_name}"

    # -------------------------------------------------------- model lifecycle --
    def setup_device(self, device=0):
        """core_system.py:156-167 — ROCm reports as ``cuda``; no MPS / CPU path here."""
        if not torch.cuda.is_available():
            raise RuntimeError("revers-o_amd needs an MI355X (ROCm) device: the hot path has no CPU fallback")
        dev = torch.device("cuda", int(device))
        print(f"🔥 Using ROCm device: {dev} ({torch.cuda.get_device_name(dev)})")
        return dev

    def load_pe_model(self, target_model=DEFAULT_VARIANT, checkpoint=None, synthetic_seed=0):
        """core_system.py:169-203: the target model first; if it is not among the available configs, or if loading it
        FAILS (:183-191 -- here: a checkpoint that does not fit the architecture, by missing, unexpected or mis-shaped
        tensor, or an engine that cannot be created), the first available config -- whose own failure propagates, as the
        reference's second ``from_config`` would."""
        print(f"📚 Loading {target_model}...")
        configs = available_configs()
        print(f"Available PE configs: {configs}")
        checkpoint = checkpoint or os.environ.get("REVERSO_PE_CHECKPOINT")
        sd = load_state_dict(checkpoint) if checkpoint else None
        if sd is None:
            # no network here: pretrained weights (pe.CLIP.from_config(..., pretrained=True), :181)
            # must be supplied as a file; without one the tower is random-initialised.
            print("⚠️ No checkpoint given (REVERSO_PE_CHECKPOINT): using seeded random-init weights")

        def build(name):
            cfg = get_config(name)
            w = sd if sd is not None else synth_weights(cfg, seed=synthetic_seed, device=self.device)
            # LayerScale follows the checkpoint's ls_* tensors; any other unexpected `visual.*` tensor is an error (weights.py)
            return VitEngine(cfg, w, device=self.device.index or 0, max_batch=self.max_batch)

        known = target_model in configs
        if not known:
            try:
                get_config(target_model)            # the build's own test miniatures are not "available configs"
                known = True
            except KeyError:
                pass
        if known:
            try:
                engine = build(target_model)
                print(f"✅ Loaded {target_model}" + (f" weights from {checkpoint}" if checkpoint else ""))
            except Exception as e:
                print(f"❌ Failed to load {target_model}: {e}")
                engine = build(configs[0])
                print(f"🔄 Using fallback: {configs[0]}")
        else:
            engine = build(configs[0])
            print(f"🔄 Using available: {configs[0]}")
        size = engine.cfg.image_size
        print("⚡ bf16 MFMA matmuls, fp32 residual stream" + (", LayerScale from the checkpoint" if engine.cfg.use_ls else ""))
        return engine, (lambda image: pp.resize_u8(image, size))

    # ------------------------------------------------------------- detection --
    def detect_regions(self, image, text_prompt=None):
        """core_system.py:237-318.  Delegates to the injected detector; without one the
        whole frame is the single region."""
        self.detected_regions = []
        self.region_embeddings = None
        self.query_embedding_for_search = None
        pil = pp.to_pil(image)
        if self.detector is not None:
            det = self.detector(pil, text_prompt or "object")
            self.detected_regions = det
            return len(det)
        w, h = pil.size
        self.detected_regions = Regions([[0, 0, w, h]], mask=None, class_names=["full_image"])
        return 1

    def _region_metadata(self, pil, regions):
        """Per-region metadata exactly as core_system.py:363-425 builds it (cap 50, empty masks
        skipped, missing masks -> full-image bbox); returns (kept indices, metadata)."""
        names = getattr(regions, "class_names", None) or ["object"]
        kept, metas = [], []
        for i in range(min(len(regions), 50)):
            conf = float(regions.confidence[i]) if i < len(regions.confidence) else 0.0
            cid = int(regions.class_id[i]) if i < len(regions.class_id) else -1
            mask = regions.mask[i] if getattr(regions, "mask", None) is not None and i < len(regions.mask) else None
            if mask is None:
                metas.append({"region_id": _uuid4(), "bbox": [0, 0, pil.width, pil.height], "area_ratio": 1.0,
                              "detection_index": i, "confidence": conf,
                              "detected_class": names[cid] if 0 <= cid < len(names) else "unknown",
                              "mask_status": "missing_or_unavailable"})
                kept.append(i)
                continue
            m = np.asarray(mask)
            m = (m > 0.5).astype(np.uint8) if np.issubdtype(m.dtype, np.floating) else m.astype(np.uint8)
            if m.sum() == 0:
                print(f"⚠️ Empty mask for region {i}, skipping")
                continue
            ys, xs = np.where(m)
            metas.append({"region_id": _uuid4(),
                          "bbox": [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())],
                          "area_ratio": float(m.sum() / m.size), "detection_index": i, "confidence": conf,
                          "detected_class": names[cid] if 0 <= cid < len(names) else "object",
                          "mask_status": "processed"})
            kept.append(i)
        return kept, metas

    # ----------------------------------------------------------------- embed --
    def _embed_pils(self, pils):
        """list of PIL images -> fp32 CPU tensor [n, D] (L2-normalised)."""
        size = self.pe_model.cfg.image_size
        if self.device_resize:
            frames = [torch.from_numpy(np.array(pp.to_pil(im), dtype=np.uint8)).to(self.device, non_blocking=True)
                      for im in pils]
            with self._lock:
                emb = self.pe_model.embed(pp.crop_resize_device(frames, None, size))
            return emb.cpu()
        u8 = torch.stack(list(self._decode_pool.map(lambda im: pp.resize_u8(im, size), pils)))
        with self._lock:
            emb = self.pe_model.embed(u8.to(self.device, non_blocking=True))
        return emb.cpu()

    def _embed_regions_batch(self, items, to_host=True, device_frames=None):
        """items: [(pil, metas)].  One vector per region of every image: the frames go to the device once,
        every region's bbox is cropped + squash-resized there in one launch (bit-identical to PIL
        crop().resize()), then forwards of max_batch crops.  Mask-derived boxes are inclusive
        (core_system.py:411).  Returns fp32 [n_regions, D] in item order: a CPU tensor, or with
        to_host=False the device tensor (the launches are asynchronous)."""
        frames, boxes = [], []
        for n_item, (pil, metas) in enumerate(items):
            if not metas:
                continue
            fi = len(frames)
            if device_frames is not None:
                frames.append(device_frames[n_item])                    # already uploaded (create_database: one copy per batch)
            else:
                frames.append(torch.from_numpy(np.array(pil, dtype=np.uint8)).to(self.device, non_blocking=True))
            for m in metas:
                x0, y0, x1, y1 = m["bbox"]
                if m.get("mask_status") == "processed":
                    x1, y1 = x1 + 1, y1 + 1
                boxes.append((fi,) + pp.clamp_box((x0, y0, x1, y1), pil.width, pil.height))
        if not boxes:
            return torch.empty((0, self.pe_model.cfg.out_dim), device=None if to_host else self.device)
        with self._lock:
            u8 = pp.crop_resize_device(frames, boxes, self.pe_model.cfg.image_size)
            emb = self.pe_model.embed(u8)
        return emb.cpu() if to_host else emb

    def _embed_regions(self, pil, metas):
        """One vector per region of one image (see _embed_regions_batch)."""
        return self._embed_regions_batch([(pil, metas)])

    def extract_embeddings(self, image):
        """core_system.py:320-429: one forward of the full image, every kept region receives
        the global embedding (:406) with its own mask-derived metadata.  With
        ``region_mode="crop"`` every region is embedded from its own crop instead."""
        if not self.detected_regions or len(self.detected_regions) == 0:
            print("❌ No regions detected")
            return [], []
        pil = pp.to_pil(image)
        kept, metas = self._region_metadata(pil, self.detected_regions)
        if self.region_mode == "crop":
            embeddings = list(self._embed_regions(pil, metas))
        else:
            g = self._embed_pils([pil])[0]
            embeddings = [g.clone() for _ in kept]
        self.region_embeddings = embeddings
        print(f"🎯 Extracted {len(embeddings)} region embeddings")
        return embeddings, metas

    def process_image_direct_pe(self, image):
        """core_system.py:431-455"""
        pil = pp.to_pil(image)
        e = self._embed_pils([pil])[0]
        if e.dim() != 1:
            raise ValueError(f"Unexpected feature shape: {tuple(e.shape)}")
        self.region_embeddings = [e]
        meta = {"region_id": _uuid4(), "bbox": [0, 0, pil.width, pil.height], "area_ratio": 1.0,
                "detection_index": 0, "confidence": 1.0, "detected_class": "full_image"}
        return [e.clone()], [meta]

    def request_stop(self):
        """core_system.py:457-459"""
        self._stop_requested = True

    # ------------------------------------------------------- gallery build ----
    def create_database(self, folder_path, database_name, text_prompt="person . car . building", use_direct_pe=False,
                        progress_callback=None, resume_from_checkpoint=False, include_subfolders=False):
        """core_system.py:461-648, with batched embedding and a working checkpoint."""
        # while this call builds <name>.building, a list_databases / load_database from another thread (a UI refresh) must
        # not adopt that directory as a crashed build's left-over (store.recover): the call swaps it in itself
        with self._building_lock:
            mine = database_name not in self._building_names      # (a second overlapping build of the SAME name: the first
            self._building_names.add(database_name)               #  one to enter keeps the entry until it returns)
        try:
            return self._create_database(folder_path, database_name, text_prompt, use_direct_pe, progress_callback,
                                         resume_from_checkpoint, include_subfolders)
        finally:
            if mine:
                with self._building_lock:
                    self._building_names.discard(database_name)

    def _create_database(self, folder_path, database_name, text_prompt, use_direct_pe, progress_callback,
                         resume_from_checkpoint, include_subfolders):
        status_messages = []

        class _Log(str):
            """What log_status returns: the whole log so far, joined only if somebody looks at it
            (the reference joins on every call, which is quadratic in the number of images)."""
            def __new__(cls, parts):
                o = super().__new__(cls, "")
                o._parts = parts
                return o

            def __str__(self):
                return "\n".join(self._parts)

        def log_status(message, progress_value=None):
            status_messages.append(message)
            if progress_callback:
                progress_callback(message, progress_value)
            return _Log(status_messages)

        os.makedirs(self.db_root, exist_ok=True)
        db_path = os.path.join(self.db_root, database_name)
        ckpt_base = os.path.join(self.db_root, "checkpoints", f"{database_name}_checkpoint")
        processed_files = set()
        # The collection is built in <db>.building (the finished one, if any, stays searchable until the new one is
        # complete) and its manifest of delta shards is the checkpoint: a resume re-opens it.
        self._partial_embeddings, self._partial_metadata = [], []          # (kept for callers that look at them: always empty now)
        build_path = db_path + BUILDING
        self._build_store = None
        build_info = self._build_info(folder_path, use_direct_pe, include_subfolders)
        if resume_from_checkpoint and os.path.exists(os.path.join(build_path, st.MANIFEST)):
            try:
                header = st.read_manifest(os.path.join(build_path, st.MANIFEST))[0]
                diff = [k for k in build_info if header.get("build", {}).get(k) != build_info[k]]
                if header.get("dim") != self.pe_model.cfg.out_dim:
                    diff.append("dim")
                if diff:
                    # another folder, model or mode: its rows must not be mixed with this build's
                    raise ValueError(f"the unfinished build was made with a different {', '.join(diff)}")
                n_files = len(os.listdir(folder_path)) if os.path.isdir(folder_path) else 0
                self._build_store = st.GalleryStore.load(build_path, device=self.device.index or 0, allow_partial=True,
                                                         capacity=n_files)        # room for the rest: no regrow right away
                processed_files = set(self._build_store.files_done)
                log_status(f"📋 Resuming from checkpoint: {len(processed_files)} files already processed")
            except Exception as e:
                log_status(f"⚠️ Error loading checkpoint: {str(e)}. Starting fresh.")
                processed_files, self._build_store = set(), None
        try:
            return self._create_database_body(folder_path, database_name, text_prompt, use_direct_pe, resume_from_checkpoint,
                                              include_subfolders, log_status, status_messages, db_path, ckpt_base,
                                              processed_files)
        finally:
            self._stop_requested = False
            self._partial_embeddings = []
            self._partial_metadata = []
            if self._build_store is not None:               # stopped or failed: the shards on disk are the checkpoint
                self._build_store.close()
                self._build_store = None

    def _build_info(self, folder_path, use_direct_pe, include_subfolders):
        """What a collection is made from and how: written into the build's manifest header, compared on resume."""
        return {"folder_path": os.path.abspath(folder_path), "model": self.pe_model.cfg.name,
                "use_ls": bool(self.pe_model.cfg.use_ls), "region_mode": self.region_mode,
                "use_direct_pe": bool(use_direct_pe), "include_subfolders": bool(include_subfolders)}

    def _create_database_body(self, folder_path, database_name, text_prompt, use_direct_pe, resume_from_checkpoint,
                              include_subfolders, log_status, status_messages, db_path, ckpt_base, processed_files):
        log_status(f"📁 Creating database '{database_name}' from {folder_path}")
        image_files = []
        if include_subfolders:
            for root, _, files in os.walk(folder_path):
                image_files += [os.path.join(root, f) for f in files
                                if any(f.lower().endswith(e) for e in IMAGE_EXTENSIONS)]
        else:
            image_files = [os.path.join(folder_path, f) for f in os.listdir(folder_path)
                           if any(f.lower().endswith(e) for e in IMAGE_EXTENSIONS)]
        image_files.sort()
        if not image_files:
            return str(log_status(f"❌ No images found in {folder_path}"))
        if resume_from_checkpoint:
            image_files = [f for f in image_files if f not in processed_files]
            if not image_files and self._build_store is None:
                return str(log_status("✅ All files already processed. Database is complete."))
            if not image_files:
                # every file was embedded before the stop / crash, but the collection was never completed: finish it
                log_status(f"📋 All files already embedded: storing {len(self._build_store)} vectors from the checkpoint")
        if image_files:
            log_status(f"📊 Found {len(image_files)} images to process", 0.1)
        if include_subfolders:
            log_status("📂 Including images from subfolders")
        log_status(f"🔧 Processing mode: {'Direct PE' if use_direct_pe else 'Detector + PE'}")
        log_status(f"📂 Database will be stored at: {db_path}")

        processed = failed = 0

        build_path = db_path + BUILDING
        collection_name = f"simple_reverso_{database_name}"
        if self._build_store is None:
            if os.path.isdir(build_path):
                shutil.rmtree(build_path)
            self._build_store = st.GalleryStore(self.pe_model.cfg.out_dim, device=self.device.index or 0,
                                                capacity=max(len(image_files), 1024), collection=collection_name,
                                                path=build_path,
                                                build_info=self._build_info(folder_path, use_direct_pe, include_subfolders))
        store_db = self._build_store

        def checkpoint():
            """the rows and finished files since the last one, as a delta shard of the collection being built; the small
            JSON next to the reference's checkpoint path only says where the build lives"""
            try:
                t_f = time.perf_counter()
                rows = store_db.flush()
                stats["flush_s"] += time.perf_counter() - t_f
                os.makedirs(os.path.dirname(ckpt_base), exist_ok=True)
                with open(ckpt_base + ".json.tmp", "w") as f:
                    json.dump({"database_name": database_name, "folder_path": folder_path, "build_path": build_path,
                               "timestamp": datetime.now().isoformat(), "processed_files": len(store_db.files_done),
                               "n_embeddings": len(store_db)}, f, indent=2)
                os.replace(ckpt_base + ".json.tmp", ckpt_base + ".json")
                if rows:
                    log_status(f"💾 Checkpoint: shard {store_db._shards - 1} ({rows} vectors, {len(store_db)} in all)")
            except Exception as e:
                log_status(f"⚠️ Error saving checkpoint: {str(e)}")

        host_resize = not self.device_resize and not (self.region_mode == "crop" and not use_direct_pe)
        model_size = self.pe_model.cfg.image_size

        stats = {"images": len(image_files), "decode_thread_s": 0.0, "wait_decode_s": 0.0, "wait_device_s": 0.0, "launch_s": 0.0,
                 "bookkeeping_s": 0.0, "flush_s": 0.0}
        self.last_ingest_stats = stats
        t_start = time.perf_counter()

        def open_rgb(path, slot=None):
            """pool task: decode (and, on the host-resize path, squash-resize) one file; with `slot` (a row of the
            pinned staging batch) the resized image is written there instead of being returned"""
            t_dec = time.perf_counter()
            try:
                return open_rgb_inner(path, slot)
            finally:
                stats["decode_thread_s"] += time.perf_counter() - t_dec      # (summed over the pool's threads; a float += under the GIL)

        def open_rgb_inner(path, slot):
            try:
                im = Image.open(path).convert("RGB")
                if not host_resize:
                    # the decoded frame goes to the device as it is (resized / cropped there): this thread leaves it in
                    # a pinned buffer of its batch slot, so that the upload is one asynchronous copy
                    return im, frame_to_pinned(im, slot)
                u8 = pp.resize_u8(im, model_size)
                if slot is None:
                    return im, u8
                slot.copy_(u8)
                return im, True
            except Exception as e:           # per-image failure: logged and skipped (core_system.py:585-591)
                return e, None

        class FrameSlab:
            """Pinned host memory for the decoded frames of ONE batch: the decode threads reserve ranges (a bump counter
            under a lock) and copy their frames in; the ingest thread uploads the used prefix with one asynchronous
            copy.  A frame that does not fit goes up from pageable memory and the slab is enlarged for the next round."""

            def __init__(self):
                # no memory yet: the first batch of a slab goes up from pageable memory (a blocking copy, but the device
                # is idle then) and tells how much the slab needs; pinning it (tens of ms) happens when the slab comes
                # round again, by which time the device has batches queued
                self.buf = torch.empty(0, dtype=torch.uint8)
                self.lock = threading.Lock()
                self.used = 0
                self.wanted = 0

            def reset(self):
                if self.wanted > self.buf.numel():
                    self.buf = torch.empty(int(self.wanted * 1.25), dtype=torch.uint8).pin_memory()
                self.used = self.wanted = 0

            def put(self, arr):
                size = (arr.size + 255) & ~255
                with self.lock:
                    off = self.used
                    self.wanted += size
                    if off + size > self.buf.numel():
                        return None, torch.from_numpy(np.array(arr, dtype=np.uint8))
                    self.used = off + size
                np.copyto(self.buf[off:off + arr.size].view(arr.shape).numpy(), arr)
                return off, arr.shape

        def frame_to_pinned(im, slot):
            arr = np.asarray(im)
            if slot is None:
                return None, torch.from_numpy(np.array(arr, dtype=np.uint8))
            return slot.put(arr)

        def upload_frames(slab, handles):
            """handles: what frame_to_pinned returned for the batch's good files.  One asynchronous copy of the slab's
            used prefix; returns the device frames (views into it), in order."""
            dev_slab = slab.buf[:slab.used].to(self.device, non_blocking=True) if slab.used else None
            out = []
            for off, x in handles:
                if off is None:
                    out.append(x.to(self.device, non_blocking=True))            # did not fit: pageable upload
                else:
                    n = x[0] * x[1] * x[2]
                    out.append(dev_slab[off:off + n].view(x))
            return out

        B = self.max_batch
        # enough decode threads to feed the device in this mode and no more (see __init__): host resize costs 9 ms of
        # thread time per 640 x 480 JPEG, decode alone 4 ms; crops are embedded three to an image
        want_threads = self.decode_workers or min(os.cpu_count() or 8,
                                                  16 if host_resize else (6 if self.region_mode == "crop" and not use_direct_pe else 8))
        if want_threads != self._decode_pool._max_workers:
            self._decode_pool.shutdown(wait=True)
            self._decode_pool = ThreadPoolExecutor(max_workers=want_threads)

        # The pool works DEPTH batches ahead of the one being embedded (one batch ahead left it idle between
        # submissions: 1 650 decoded images/s in the loop against 3 700 for the pool alone).  Batch n's host buffer --
        # staging rows or frame slab -- is number n % NSLOT: decoding (DEPTH of them) / waiting / being uploaded.
        DEPTH = 2
        NSLOT = DEPTH + 2
        slabs = [FrameSlab() for _ in range(NSLOT)] if not host_resize else None
        frames_done = [None] * NSLOT         # event behind the upload that read a slab

        def submit(s0):
            if stage is not None:            # batch s0 // B fills staging buffer (s0 // B) & 1, one row per file
                buf = stage[(s0 // B) % NSLOT]
                return [self._decode_pool.submit(open_rgb, p, buf[j]) for j, p in enumerate(image_files[s0:s0 + B])]
            if not host_resize:
                slab = slabs[(s0 // B) % NSLOT]
                slab.reset()
                return [self._decode_pool.submit(open_rgb, p, slab) for p in image_files[s0:s0 + B]]
            return [self._decode_pool.submit(open_rgb, p) for p in image_files[s0:s0 + B]]

        # Three things overlap: the pool decodes batch i+1, the device embeds batch i (its vectors come back through a
        # pinned buffer behind an event), and this thread does the per-image bookkeeping of batch i-1.
        last_ckpt = time.monotonic()
        crop_mode = self.region_mode == "crop" and not use_direct_pe
        pipelined = host_resize and not crop_mode
        stage = None
        if pipelined:
            stage = [torch.zeros((B, 3, model_size, model_size), dtype=torch.uint8).pin_memory() for _ in range(NSLOT)]
        h2d_done = [None] * NSLOT
        pending = []                         # futures of the batches being decoded, oldest first
        submitted = [0]

        def top_up(it):
            """keep the pool DEPTH batches ahead of batch `it`; a buffer is handed out again only after the upload that
            read it (NSLOT batches ago) has left the host"""
            while submitted[0] <= it + DEPTH and submitted[0] * B < len(image_files):
                bn = submitted[0]
                t_ev = time.perf_counter()
                for ev in (h2d_done[bn % NSLOT], frames_done[bn % NSLOT]):
                    if ev is not None:
                        ev.synchronize()
                stats["wait_device_s"] += time.perf_counter() - t_ev        # the device is the bottleneck while this grows
                pending.append(submit(bn * B))
                submitted[0] += 1

        def detect_batch(pils):
            """crop mode, as soon as a batch is decoded: detector boxes and region metadata per image (None for a file
            that failed to decode), so that the batch's crops can be queued on the device before the bookkeeping of the
            batch in front of it"""
            det = []
            for im in pils:
                if isinstance(im, Exception):
                    det.append(None)
                    continue
                n_reg = self.detect_regions(im, text_prompt)
                det.append((n_reg, self._region_metadata(im, self.detected_regions)[1] if n_reg else []))
            return det

        def finalize(item):
            """bookkeeping of one embedded batch: metadata per image, then the batch's vectors -- still on the device --
            go straight into the collection being built (device-to-device append; nothing visits the host)"""
            nonlocal failed
            s, paths, pils, hosts, dev_emb, by_file, last, det = item
            gi = 0
            batch_items = []                       # (path, pil, metas, row of dev_emb or None)
            for j, (path, im) in enumerate(zip(paths, pils)):
                i = s + j
                filename = os.path.basename(path)
                log_status(f"🔄 Processing {i + 1}/{len(image_files)}: {filename}", 0.1 + 0.7 * (i / len(image_files)))
                if isinstance(im, Exception):
                    log_status(f"❌ Error processing {filename}: {str(im)}")
                    done_files.append(path)
                    failed += 1
                    continue
                row = None
                if dev_emb is not None and not crop_mode:
                    row = j if by_file else gi                # staged batches have one row per file, the others one per decoded file
                    gi += 1
                if use_direct_pe:
                    metas = [{"region_id": _uuid4(), "bbox": [0, 0, im.width, im.height], "area_ratio": 1.0,
                              "detection_index": 0, "confidence": 1.0, "detected_class": "full_image"}]
                    log_status(f"✅ Extracted global embedding for {filename}")
                else:
                    if det is not None:
                        n_reg, metas = det[j]                 # crop mode: detected when the batch was queued
                    else:
                        n_reg = self.detect_regions(im, text_prompt)
                        metas = self._region_metadata(im, self.detected_regions)[1] if n_reg else []
                    if n_reg == 0:
                        log_status(f"⚠️ No regions found in {filename}, skipping")
                        done_files.append(path)
                        failed += 1
                        continue
                    log_status(f"✅ Found {n_reg} regions, extracted {len(metas)} embeddings in {filename}")
                for m in metas:
                    m["image_source"] = path
                    m["filename"] = filename
                    m["original_region_id"] = m.get("region_id", _uuid4())
                    m["region_id"] = _uuid4()
                batch_items.append((path, im, metas, row, hosts[j]))
            if crop_mode:
                store(batch_items, dev_emb, None, last)       # one vector per region, in item order (queued with the batch)
                return
            store(batch_items, None, dev_emb, last)

        done_files = []                            # files finished (stored, failed or empty) since the last upsert

        def store(batch_items, region_vecs, dev_emb, last):
            """the embedded batch enters the collection (and only now counts as processed: a shard never names a file
            whose vectors it does not hold)"""
            nonlocal processed, last_ckpt
            metas_all, rows = [], []
            for path, im, metas, row, _ in batch_items:
                metas_all.extend(metas)
                if region_vecs is None:
                    rows.extend([row] * len(metas))            # every region of an image stores its global vector (core_system.py:406-408)
                done_files.append(path)
                processed += 1
                self._last_processed_file = path
            if metas_all:
                with self._lock, torch.cuda.device(self.device):
                    if region_vecs is not None:
                        vec = region_vecs
                    elif rows == list(range(len(rows))) and len(rows) == dev_emb.shape[0]:
                        vec = dev_emb
                    else:
                        vec = dev_emb.index_select(0, torch.tensor(rows, dtype=torch.int64, device=dev_emb.device))
                    store_db.upsert(vec, [m["region_id"] for m in metas_all], metas_all, files=list(done_files))
            else:
                store_db.upsert(torch.zeros((0, store_db.dim)), [], [], files=list(done_files))
            processed_files.update(done_files)
            done_files.clear()
            # a checkpoint writes only what is new: at most one per interval (the end of the build writes the last shard itself)
            if time.monotonic() - last_ckpt >= self.checkpoint_interval_s and not last:
                checkpoint()
                last_ckpt = time.monotonic()

        inflight = None
        for it, s in enumerate(range(0, len(image_files), B)):
            if self._stop_requested:
                if inflight is not None:
                    finalize(inflight)
                    inflight = None
                log_status("🛑 Stop requested. Saving progress...")
                checkpoint()
                return "\n".join(status_messages) + "\n\n⏸️ Processing stopped. You can resume later."
            paths = image_files[s:s + B]
            top_up(it)
            futures = pending.pop(0)
            t_w = time.perf_counter()
            results = [f.result() for f in futures]
            stats["wait_decode_s"] += time.perf_counter() - t_w
            t_l = time.perf_counter()
            pils = [r[0] for r in results]
            hosts = [r[1] for r in results]
            good = [r for r in results if not isinstance(r[0], Exception)]
            # global vectors of the whole batch in one forward -- or, when every region is cropped, the regions' vectors
            dev_emb = None
            det = None
            if crop_mode:
                # region crops of the whole batch: boxes from the detector, the frames go up from their pinned slab (one
                # asynchronous copy), one crop + resize launch, forwards of max_batch crops -- queued NOW, so that the
                # device holds this batch while the thread does the bookkeeping of the one before; the vectors stay there
                det = detect_batch(pils)
                with_regions = [j for j, d in enumerate(det) if d is not None and d[1]]
                if with_regions:
                    with torch.cuda.device(self.device):
                        dev_emb = self._embed_regions_batch([(pils[j], det[j][1]) for j in with_regions], to_host=False,
                                                            device_frames=upload_frames(slabs[it % NSLOT],
                                                                                        [hosts[j] for j in with_regions]))
                        frames_done[it % NSLOT] = torch.cuda.Event()
                        frames_done[it % NSLOT].record()
            elif good:
                if pipelined:
                    # every file of the batch has its row in the staging buffer (a failed file's row keeps whatever it
                    # held: embedded and never looked at)
                    buf = stage[it % NSLOT][:len(paths)]
                    with self._lock, torch.cuda.device(self.device):
                        dev_in = buf.to(self.device, non_blocking=True)
                        h2d_done[it % NSLOT] = torch.cuda.Event()
                        h2d_done[it % NSLOT].record()
                        dev_emb = self.pe_model.embed(dev_in)
                else:
                    # device resize: the decoded frames go up as they are, one crop + resize launch, one forward
                    with self._lock, torch.cuda.device(self.device):
                        frames = upload_frames(slabs[it % NSLOT], [h for _, h in good])
                        frames_done[it % NSLOT] = torch.cuda.Event()
                        frames_done[it % NSLOT].record()
                        dev_emb = self.pe_model.embed(pp.crop_resize_device(frames, None, model_size))
            stats["launch_s"] += time.perf_counter() - t_l
            cur = (s, paths, pils, hosts, dev_emb, pipelined, s + B >= len(image_files), det)
            if inflight is not None:
                t_b = time.perf_counter()
                finalize(inflight)
                stats["bookkeeping_s"] += time.perf_counter() - t_b
            inflight = cur
        if inflight is not None:
            t_b = time.perf_counter()
            finalize(inflight)
            stats["bookkeeping_s"] += time.perf_counter() - t_b

        if len(store_db) == 0:
            store_db.close()                                    # nothing to keep: no empty build directory or note stays behind
            self._build_store = None
            shutil.rmtree(build_path, ignore_errors=True)
            st.remove_checkpoint(ckpt_base)
            return str(log_status("❌ No embeddings extracted from any images"))

        log_status(f"📦 Recreated collection: {collection_name}", 0.8)
        if self._stop_requested:
            # a stop between the last embed batch and the completion of the collection: everything is in the shards
            log_status("🛑 Stop requested during database storage. Progress saved.")
            checkpoint()
            return "\n".join(status_messages) + "\n\n⏸️ Processing stopped. You can resume later."
        with self._lock:
            t_f = time.perf_counter()
            store_db.save()                                     # the last delta shard + the "complete" line
            stats["flush_s"] += time.perf_counter() - t_f
            log_status(f"💾 Stored {len(store_db)} points in {store_db._shards} shards", 0.9)
            if self.vector_db is not None:
                self.vector_db.close()
            st.swap_in(build_path, db_path)                     # recreate_collection: the new one replaces the old one only now
            store_db.path = db_path
            self.vector_db = store_db
            self._build_store = None
            self.current_database = collection_name
        if os.path.exists(ckpt_base + ".json"):
            st.remove_checkpoint(ckpt_base)
            log_status("🧹 Cleaned up checkpoint file")
        stats["total_s"] = time.perf_counter() - t_start
        stats["vectors"] = len(self.vector_db)
        log_status("\n📊 Final Summary:", 0.9)
        log_status(f"✅ Successfully processed: {processed} images")
        if failed > 0:
            log_status(f"⚠️ Failed to process: {failed} images")
        log_status(f"🔍 Total embeddings stored: {len(self.vector_db)}")
        log_status(f"🎯 Database '{database_name}' ready for searching!", 1.0)
        return "\n".join(status_messages)

    # ---------------------------------------------------------------- search --
    def search_similar(self, similarity_threshold=0.7, max_results=5):
        """core_system.py:650-717: first region embedding as the query, limit + score
        threshold, results best first with the same text and thumbnail formatting."""
        if not self.region_embeddings:
            return "❌ No query embeddings available. Please detect/process an image first.", []
        if not self.vector_db or not self.current_database:
            return "❌ No database loaded. Please create or load a database first.", []
        query = self.region_embeddings[0]
        with self._lock:
            hits = self.vector_db.search(query, limit=int(max_results), score_threshold=float(similarity_threshold))
        if not hits:
            return f"❌ No similar regions found above threshold {similarity_threshold}", []
        text = f"🎯 Found {len(hits)} similar regions:\n\n"

        def thumbnail(r):
            """the hit's source image with its score label, at most 400 px (core_system.py:682-706); None when the file
            is gone or unreadable.  One pool task per hit: decoding and resampling release the GIL, and ten hits were
            40 ms of a 50 ms interactive query when done one after the other."""
            image_path = r.payload.get("image_source", "")
            if not os.path.exists(image_path):
                return None
            try:
                img = Image.open(image_path).convert("RGB")
                draw = ImageDraw.Draw(img)
                try:
                    font = ImageFont.truetype("Arial.ttf", max(15, int(min(img.height, img.width) * 0.05)))
                except IOError:
                    font = ImageFont.load_default()
                label = f"Score: {r.score:.3f}"
                box = draw.textbbox((5, 5), label, font=font)
                draw.rectangle([box[0] - 2, box[1] - 2, box[2] + 2, box[3] + 2], fill="black")
                draw.text((5, 5), label, fill="white", font=font)
                img.thumbnail((400, 400), Image.Resampling.LANCZOS)
                return img
            except Exception as e:
                print(f"❌ Error loading/processing image {image_path}: {e}")
                return None

        try:
            images = list(self._decode_pool.map(thumbnail, hits)) if len(hits) > 1 else [thumbnail(hits[0])]
        except RuntimeError:                     # the pool is being replaced by a build on another thread
            images = [thumbnail(r) for r in hits]
        items = []
        for i, (r, img) in enumerate(zip(hits, images)):
            payload = r.payload
            filename = payload.get("filename", "Unknown")
            image_path = payload.get("image_source", "")
            text += f"{i + 1}. {filename} (Similarity: {r.score:.3f})\n"
            text += f"   Source: {image_path}\n"
            text += f"   📍 Bounding box: {str(payload.get('bbox', '[0,0,0,0]'))}\n\n"
            items.append({"image": img, "score": r.score, "filename": filename, "bbox": payload.get("bbox")})
        return text, items

    def visualize_detections(self, image, selected_region_index=None):
        """core_system.py:719-757 draws mask contours with OpenCV (UI cosmetics, out of scope);
        boxes are outlined with PIL so the UI keeps working."""
        pil = pp.to_pil(image).copy()
        if not self.detected_regions or len(self.detected_regions) == 0:
            return pil
        draw = ImageDraw.Draw(pil)
        for i, box in enumerate(self.detected_regions.xyxy):
            sel = i == selected_region_index
            draw.rectangle([float(v) for v in box], outline=(0, 255, 0) if sel else (255, 0, 0), width=3 if sel else 1)
            draw.text((float(box[0]) + 3, float(box[1]) + 3), f"{i + 1}", fill=(0, 0, 0) if sel else (255, 0, 0))
        return pil
