"""Host side of the hot path: ``embed()`` / ``search()`` over librevo's C ABI.

Python owns orchestration only (device tensors, streams, weight hand-over); every
kernel on the path is HIP inside ``librevo.so``.  The entry points mirror what
the reference does at

* ``core_system.py:431-455`` ``process_image_direct_pe``  ->  :meth:`VitEngine.embed`
* ``core_system.py:596-622`` collection create + upsert     ->  :class:`Gallery`
* ``core_system.py:650-664`` ``search_similar`` / ``vector_db.search``  ->  :meth:`Gallery.search`
"""
import ctypes as C
import json
import threading

import torch

from . import _lib
from .config import PEConfig, get_config
from .weights import check_state_dict, resolve_config, strip_non_parameters, synth_weights

IMAGE_F32 = 0
IMAGE_U8 = 1


def _require_cuda(t, name, device=None):
    if not t.is_cuda:
        raise _lib.RevoError(f"{name} must be a device tensor (the hot path has no CPU fallback)")
    if device is not None and t.device != device:
        raise _lib.RevoError(f"{name} is on {t.device} but the handle is bound to {device}")


def _as_device(device):
    d = device if isinstance(device, torch.device) else torch.device("cuda", int(device))
    return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())


class VitEngine:
    """A PE vision tower resident on one GPU (weights bf16, fp32 residual stream)."""

    def __init__(self, cfg: PEConfig, state_dict, device=0, max_batch=64, experiments=False):
        # LayerScale follows the checkpoint; unknown `visual.*` tensors are rejected by name
        cfg = resolve_config(cfg, state_dict)
        self.cfg = cfg
        self.device = _as_device(device)
        self.max_batch = int(max_batch)
        self._lock = threading.Lock()
        # experiments=True: the handle lives in librevo_exp.so, the only library with the parity-test hooks
        # (residual_after / taps); the product library cannot stop a forward early or hand out intermediates
        self.experiments = bool(experiments) or _lib.product_is_experiment_build()
        lib = _lib.load_exp() if experiments else _lib.load()
        check_state_dict(cfg, state_dict)
        # the image tower only (a whole-CLIP state dict also holds the text tower), without registered buffers
        state_dict = strip_non_parameters({k: v for k, v in state_dict.items() if k.startswith("visual.")})
        names = sorted(state_dict)
        keep = []          # keep converted tensors alive until create returns
        arr = (_lib.Tensor * len(names))()
        for i, n in enumerate(names):
            t = state_dict[n].detach().to(torch.float32).contiguous()
            keep.append(t)
            arr[i].name = n.encode()
            arr[i].data = t.data_ptr()
            arr[i].numel = t.numel()
        c = _lib.VitCfg(cfg.image_size, cfg.patch_size, cfg.width, cfg.layers, cfg.heads, cfg.mlp_dim, cfg.out_dim,
                        cfg.pool_heads, int(cfg.use_cls), int(cfg.use_ls), cfg.ln_eps, cfg.rope_theta, cfg.pool_mlp_dim)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _lib.check(lib.revo_vit_create(C.byref(c), arr, len(names), self.device.index or 0, self.max_batch,
                                           C.byref(h)), "revo_vit_create")
        self._h = h
        self._lib = lib

    @classmethod
    def synthetic(cls, name_or_cfg="PE-Core-L14-336", seed=0, device=0, max_batch=64, experiments=False, **kw):
        cfg = name_or_cfg if isinstance(name_or_cfg, PEConfig) else get_config(name_or_cfg)
        dev = torch.device("cuda", device)
        sd = synth_weights(cfg, seed=seed, device=dev, **kw)
        return cls(cfg, sd, device=device, max_batch=max_batch, experiments=experiments)

    def _need_hooks(self, what):
        if not self.experiments:
            raise _lib.RevoError(f"{what} is a parity-test hook of librevo_exp.so: create the engine with experiments=True")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.revo_vit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the embed entry point ------------------------------------------------
    def embed(self, images, normalize=True, out=None):
        """images: uint8 or float32 ``[B,3,H,W]`` device tensor at the model
        resolution (float input already normalised to [-1,1], i.e. what
        ``self.preprocess`` yields at core_system.py:439).  Returns fp32
        ``[B, out_dim]`` on the device, L2-normalised (core_system.py:447)."""
        _require_cuda(images, "images", self.device)
        cfg = self.cfg
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != cfg.image_size or images.shape[3] != cfg.image_size:
            raise ValueError(f"images must be [B,3,{cfg.image_size},{cfg.image_size}], got {tuple(images.shape)}")
        if images.dtype == torch.uint8:
            kind = IMAGE_U8
        elif images.dtype == torch.float32:
            kind = IMAGE_F32
        else:
            raise ValueError("images must be uint8 or float32")
        images = images.contiguous()
        B = images.shape[0]
        if out is None:
            out = torch.empty((B, cfg.out_dim), dtype=torch.float32, device=images.device)
        with self._lock, torch.cuda.device(self.device):
            st = _lib.current_stream()
            for s in range(0, B, self.max_batch):
                e = min(B, s + self.max_batch)
                _lib.check(self._lib.revo_vit_forward(self._h, _lib.ptr(images[s:e]), kind, e - s, _lib.ptr(out[s:e]),
                                                      int(bool(normalize)), st), "revo_vit_forward")
        return out

    def ln_fold_stats(self, reset=False):
        """Telemetry of the LayerNorm folded into qkv / fc1 (include/revo.h revo_vit_stats; synchronises the current stream):
        rows whose statistics the consuming GEMMs merged since the last reset, how many of them had |mean| / std above
        ``ratio`` (and above four times it) -- the regime in which the fold's rounding error exceeds the LayerNorm kernel's."""
        out = (C.c_double * 4)()
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.revo_vit_stats(self._h, out, int(bool(reset)), _lib.current_stream()), "revo_vit_stats")
        return {"rows": int(out[0]), "rows_above_ratio": int(out[1]), "rows_above_4x_ratio": int(out[2]), "ratio": float(out[3])}

    # -- parity-test hook -----------------------------------------------------
    def residual_after(self, images, n_layers):
        """fp32 residual stream [B, S, W] after ln_pre and the first n blocks."""
        self._need_hooks("residual_after")
        _require_cuda(images, "images", self.device)
        B = images.shape[0]
        assert B <= self.max_batch
        kind = IMAGE_U8 if images.dtype == torch.uint8 else IMAGE_F32
        x = torch.empty((B, self.cfg.seq, self.cfg.width), dtype=torch.float32, device=images.device)
        dummy = torch.empty((B, self.cfg.out_dim), dtype=torch.float32, device=images.device)
        with self._lock, torch.cuda.device(self.device):
            st = _lib.current_stream()
            _lib.check(self._lib.revo_vit_set_debug_layers(self._h, int(n_layers)))
            try:
                _lib.check(self._lib.revo_vit_forward(self._h, _lib.ptr(images.contiguous()), kind, B, _lib.ptr(dummy),
                                                      0, st), "revo_vit_forward")
                _lib.check(self._lib.revo_vit_read_residual(self._h, B, _lib.ptr(x), st))
            finally:
                self._lib.revo_vit_set_debug_layers(self._h, -1)
        return x


    def taps(self, images):
        """Parity-test hook: intermediate activations of one forward as fp32 CPU-comparable tensors:
        ``embed`` [B,S,W] (patch embed + position + class token, before ln_pre), ``ln_post`` [B,S,W] (fp32: the head
        works in fp32), ``pooled`` [B,W] (attention-pool output before proj) and the ``embedding`` [B,D]."""
        self._need_hooks("taps")
        _require_cuda(images, "images", self.device)
        B = images.shape[0]
        assert B <= self.max_batch
        cfg = self.cfg
        out = {"embed": self.residual_after(images, -2)}
        emb = self.embed(images)
        lnp = torch.empty((B, cfg.seq, cfg.width), dtype=torch.float32, device=self.device)
        pooled = torch.empty((B, cfg.width), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            st = _lib.current_stream()
            _lib.check(self._lib.revo_vit_read_tap(self._h, 1, B, _lib.ptr(lnp), st), "revo_vit_read_tap")
            _lib.check(self._lib.revo_vit_read_tap(self._h, 2, B, _lib.ptr(pooled), st), "revo_vit_read_tap")
        out.update(ln_post=lnp, pooled=pooled, embedding=emb)
        return out


class Gallery:
    """Device-resident cosine gallery: normalised rows as bf16 (scan copy) plus an
    fp32 master copy used for exact re-scoring and persistence."""

    def __init__(self, dim, capacity, device=0, keep_f32=True, experiments=False):
        self.dim = int(dim)
        self.capacity = int(capacity)
        self.device = _as_device(device)
        self._lock = threading.Lock()
        self._total_rows = 0                      # set_total_rows: rows of the sharded gallery this handle is a shard of (0: none)
        # experiments=True: a handle of librevo_exp.so, the only library in which a search can be forced through (or
        # kept from) its exact fallbacks (set_search_mode: parity tests and timing scripts)
        self.experiments = bool(experiments) or _lib.product_is_experiment_build()
        self._lib = _lib.load_exp() if experiments else _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.revo_gallery_create(self.dim, self.capacity, self.device.index or 0, int(keep_f32),
                                                     C.byref(h)), "revo_gallery_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.revo_gallery_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.revo_gallery_size(self._h))

    def clear(self):
        _lib.check(self._lib.revo_gallery_clear(self._h))

    def add(self, vectors, normalize=True):
        """Append fp32 rows [n, dim] (device or host tensor).  Rows are normalised
        at insert like qdrant's COSINE collections (core_system.py:600-603)."""
        v = vectors.detach().to(torch.float32).contiguous()
        if v.dim() != 2 or v.shape[1] != self.dim:
            raise ValueError(f"vectors must be [n, {self.dim}], got {tuple(v.shape)}")
        if v.is_cuda:
            _require_cuda(v, "vectors", self.device)
        start = len(self)
        with self._lock, torch.cuda.device(self.device):
            # a device source is read by kernels on torch's current stream: its memory is not reused before they have
            # run (the caching allocator is stream-ordered), so nothing waits here -- an ingest appends batch after
            # batch without a host round trip; a host source is staged and synchronised inside the call
            _lib.check(self._lib.revo_gallery_append(self._h, _lib.ptr(v), v.shape[0], int(bool(normalize)),
                                                     int(v.is_cuda), _lib.current_stream()), "revo_gallery_append")
        return start

    def read(self, start=0, n=None):
        n = len(self) - start if n is None else n
        out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            _lib.check(self._lib.revo_gallery_read(self._h, start, n, _lib.ptr(out), 1), "revo_gallery_read")
        return out

    def search(self, queries, k=5, score_threshold=None, index_offset=0):
        """queries: fp32 [Q, dim] device tensor.  Returns (scores [Q,k] fp32,
        indices [Q,k] int64, counts [Q] int32), best first, padded with -inf/-1
        past ``counts`` (the reference's ``limit`` / ``score_threshold`` semantics,
        core_system.py:659-664)."""
        _require_cuda(queries, "queries", self.device)
        q = queries.detach().to(torch.float32).contiguous()
        if q.dim() == 1:
            q = q[None]
        if q.shape[1] != self.dim:
            raise ValueError(f"queries must be [Q, {self.dim}], got {tuple(q.shape)}")
        Q = q.shape[0]
        scores = torch.empty((Q, k), dtype=torch.float32, device=q.device)
        idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
        counts = torch.empty((Q,), dtype=torch.int32, device=q.device)
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.revo_search_topk(
                self._h, _lib.ptr(q), Q, int(k), int(score_threshold is not None),
                float(score_threshold if score_threshold is not None else 0.0), int(index_offset),
                _lib.ptr(scores), _lib.ptr(idx), _lib.ptr(counts), _lib.current_stream()), "revo_search_topk")
        return scores, idx, counts


    # -- the exactness certificate (include/revo.h, "EXACTNESS") ------------------------------------------
    MODES = {"certified": 0, "collect": 1, "bruteforce": 2, "uncertified": 3}

    def set_search_mode(self, mode="certified"):
        """``certified`` (default): every query's result is certified exact or re-done exactly; ``collect`` /
        ``bruteforce``: every query takes that fallback (parity tests); ``uncertified``: certificate counted only."""
        if not self.experiments:
            raise _lib.RevoError("set_search_mode is a parity-test hook of librevo_exp.so: create the gallery with "
                                 "experiments=True (the product library's searches are always certified exact)")
        _lib.check(self._lib.revo_search_set_mode(self._h, self.MODES[mode] if isinstance(mode, str) else int(mode)),
                   "revo_search_set_mode")

    def search_stats(self):
        """Counters of the last search on this handle (synchronises the current stream)."""
        out = (C.c_int32 * 8)()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.revo_search_stats(self._h, out, _lib.current_stream()), "revo_search_stats")
        return {"uncertified": int(out[0]), "bruteforced": int(out[1]), "checked": int(out[2]), "collected_rows": int(out[3]),
                "from_segments": int(out[4])}

    def set_total_rows(self, total_rows):
        """This handle holds ONE SHARD of a row-sharded gallery of ``total_rows`` rows (0: forget): its two-phase scans
        start from an estimate of the whole gallery's admission level (include/revo.h, revo_search_set_total_rows)."""
        _lib.check(self._lib.revo_search_set_total_rows(self._h, int(total_rows)), "revo_search_set_total_rows")
        self._total_rows = int(total_rows)

    def search_plan(self, n_queries, k=5):
        """How a search would run (reporting): dict with the scan form, the pre-pass rows, slices and ksel."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.revo_search_plan(self._h, int(n_queries), int(k), out), "revo_search_plan")
        return {"scan256": bool(out[0]), "prepass_rows": int(out[1]), "slices": int(out[2]), "ksel": int(out[3])}

    # -- the same search in two phases (row-sharded gallery; see sharded.py and include/revo.h) --------------
    def search_candidates(self, queries, k=5, top_m=8):
        """Phase 1: scan this shard, keep the candidates in the handle.  Returns int32 ``[Q, top_m]``: the bit
        patterns of the order-preserving uint32 scan scores of each query's best ``top_m`` candidates."""
        _require_cuda(queries, "queries", self.device)
        q = queries.detach().to(torch.float32).contiguous()
        if q.dim() == 1:
            q = q[None]
        if q.shape[1] != self.dim:
            raise ValueError(f"queries must be [Q, {self.dim}], got {tuple(q.shape)}")
        bounds = torch.empty((q.shape[0], int(top_m)), dtype=torch.int32, device=q.device)
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.revo_search_candidates(self._h, _lib.ptr(q), q.shape[0], int(k), int(top_m),
                                                        _lib.ptr(bounds), _lib.current_stream()), "revo_search_candidates")
        return bounds

    def search_finish(self, n_queries, k, all_bounds=None, score_threshold=None, index_offset=0, out_packed=None):
        """Phase 2: fp32 re-score of the candidates that can still be among the best of the whole gallery
        (``all_bounds``: int32 ``[parts, Q, top_m]``, the all-gathered phase-1 outputs; None = every candidate).
        Returns (scores, indices, counts); with ``out_packed`` (uint8 ``[packed_bytes(Q, k)]``) scores and indices
        are views into that buffer, laid out for :func:`merge_topk_packed`."""
        Q, k = int(n_queries), int(k)
        cert = None
        # (after a scan that dropped rows against an ESTIMATED level -- set_total_rows on a shard large enough to estimate --
        #  only the packed block's certificate makes the result exhaustive: revo_search_finish itself refuses
        #  cert = NULL then, status -2, and says so; a shard that never estimated is served either way)
        if out_packed is not None:
            idx = out_packed[: Q * k * 8].view(torch.int64).view(Q, k)
            scores = out_packed[Q * k * 8: Q * k * 12].view(torch.float32).view(Q, k)
            cert = out_packed[Q * k * 12: Q * k * 12 + Q * 4].view(torch.float32)     # this shard's certificate bounds
        else:
            scores = torch.empty((Q, k), dtype=torch.float32, device=self.device)
            idx = torch.empty((Q, k), dtype=torch.int64, device=self.device)
        counts = torch.empty((Q,), dtype=torch.int32, device=self.device)
        parts = top_m = 0
        if all_bounds is not None:
            _require_cuda(all_bounds, "all_bounds", self.device)
            all_bounds = all_bounds.contiguous()
            parts, top_m = int(all_bounds.shape[0]), int(all_bounds.shape[2])
            if all_bounds.dtype != torch.int32 or all_bounds.shape[1] != Q:
                raise ValueError("all_bounds must be int32 [parts, Q, top_m]")
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.revo_search_finish(
                self._h, Q, k, int(score_threshold is not None),
                float(score_threshold if score_threshold is not None else 0.0), int(index_offset), _lib.ptr(all_bounds),
                parts, top_m, _lib.ptr(scores), _lib.ptr(idx), _lib.ptr(counts), _lib.ptr(cert), _lib.current_stream()),
                "revo_search_finish")
        return scores, idx, counts

    def search_exact(self, q_idx, need, k, index_offset=0, out_packed=None):
        """Second round of a row-sharded search: exact local top-k of the queries ``q_idx`` (int32 ``[n]``, rows of the
        last :meth:`search_candidates` call) given ``need`` (fp32 ``[n]``: the score a row must reach to change the
        merged result).  Results in compact rows ``[n, k]``; with ``out_packed`` laid out for :func:`merge_topk_packed`."""
        n, k = int(q_idx.shape[0]), int(k)
        _require_cuda(q_idx, "q_idx", self.device)
        _require_cuda(need, "need", self.device)
        assert q_idx.dtype == torch.int32 and need.dtype == torch.float32 and need.shape[0] == n
        if out_packed is not None:
            idx = out_packed[: n * k * 8].view(torch.int64).view(n, k)
            scores = out_packed[n * k * 8: n * k * 12].view(torch.float32).view(n, k)
            out_packed[n * k * 12: n * k * 12 + n * 4].view(torch.float32).fill_(float("-inf"))   # exact: nothing left to certify
        else:
            scores = torch.empty((n, k), dtype=torch.float32, device=self.device)
            idx = torch.empty((n, k), dtype=torch.int64, device=self.device)
        counts = torch.empty((n,), dtype=torch.int32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.revo_search_exact(self._h, n, _lib.ptr(q_idx.contiguous()), _lib.ptr(need.contiguous()), k, 0,
                                                   0.0, int(index_offset), _lib.ptr(scores), _lib.ptr(idx), _lib.ptr(counts),
                                                   _lib.current_stream()), "revo_search_exact")
        return scores, idx, counts


def search_estimates(k):
    """True if shards that know the whole gallery's row count scan a best-k search against its estimated admission level."""
    return int(_lib.load().revo_search_estimates(int(k))) == 1


def search_ksel(k):
    """Candidates the scan keeps per query for a top-k search (32 or 64)."""
    return int(_lib.load().revo_search_ksel(int(k)))


def packed_bytes(n_queries, k):
    """Size of one packed result block ([Q, k] int64 indices, [Q, k] fp32 scores, [Q] fp32 certificate bounds,
    padded to 16 bytes)."""
    return int(_lib.load().revo_topk_packed_bytes(int(n_queries), int(k)))


def merge_topk_packed(packed, parts, n_queries, k, score_threshold=None, certify=False):
    """Merge ``parts`` packed result blocks (uint8, back to back: one all-gather) into [Q, k]; K14.
    ``certify``: also check every query's exactness certificate over all shards; returns a fourth item
    ``(unc_count int32 [1], unc_q int32 [Q], unc_need fp32 [Q])`` (device; the first ``unc_count`` entries are valid)."""
    _require_cuda(packed, "packed")
    Q, k = int(n_queries), int(k)
    assert packed.dtype == torch.uint8 and packed.numel() == parts * packed_bytes(Q, k)
    scores = torch.empty((Q, k), dtype=torch.float32, device=packed.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=packed.device)
    counts = torch.empty((Q,), dtype=torch.int32, device=packed.device)
    unc = None
    if certify:
        unc = (torch.zeros((1,), dtype=torch.int32, device=packed.device),
               torch.empty((max(Q, 1),), dtype=torch.int32, device=packed.device),
               torch.empty((max(Q, 1),), dtype=torch.float32, device=packed.device))
    lib = _lib.load()
    with torch.cuda.device(packed.device):
        _lib.check(lib.revo_topk_merge_packed(_lib.ptr(packed), int(parts), Q, k, int(score_threshold is not None),
                                              float(score_threshold if score_threshold is not None else 0.0),
                                              _lib.ptr(scores), _lib.ptr(idx), _lib.ptr(counts),
                                              _lib.ptr(unc[0]) if unc else None, _lib.ptr(unc[1]) if unc else None,
                                              _lib.ptr(unc[2]) if unc else None, _lib.current_stream()),
                   "revo_topk_merge_packed")
    return (scores, idx, counts, unc) if certify else (scores, idx, counts)


def merge_topk(part_scores, part_indices, k, score_threshold=None):
    """Merge per-shard results [P,Q,k] (device) into [Q,k]; K14."""
    _require_cuda(part_scores, "part_scores")
    P, Q, kk = part_scores.shape
    assert kk == k
    ps = part_scores.contiguous()
    pi = part_indices.contiguous()
    scores = torch.empty((Q, k), dtype=torch.float32, device=ps.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=ps.device)
    counts = torch.empty((Q,), dtype=torch.int32, device=ps.device)
    lib = _lib.load()
    with torch.cuda.device(ps.device):
        _lib.check(lib.revo_topk_merge(_lib.ptr(ps), _lib.ptr(pi), P, Q, k, int(score_threshold is not None),
                                       float(score_threshold if score_threshold is not None else 0.0),
                                       _lib.ptr(scores), _lib.ptr(idx), _lib.ptr(counts), _lib.current_stream()),
                   "revo_topk_merge")
    return scores, idx, counts


# ---- module-level convenience API named by the north star ------------------
_default_engine = None
_default_gallery = None


def set_default(engine=None, gallery=None):
    global _default_engine, _default_gallery
    if engine is not None:
        _default_engine = engine
    if gallery is not None:
        _default_gallery = gallery


def embed(images, engine=None):
    """images: uint8 / float32 ``[B,3,H,W]`` device tensor at the model resolution, or a list of PIL images / arrays /
    paths (squash-resized on the host like ``self.preprocess`` at core_system.py:439, uploaded as uint8).
    Returns fp32 ``[B, D]`` on the device, L2-normalised."""
    eng = engine or _default_engine
    if eng is None:
        raise _lib.RevoError("embed(): no engine; create a VitEngine and call set_default(engine=...)")
    if isinstance(images, (list, tuple)):
        from . import preprocess as pp
        images = pp.batch_u8(list(images), eng.cfg.image_size, pin=True).to(eng.device, non_blocking=True)
    return eng.embed(images)


def search(queries, k=5, score_threshold=None, gallery=None):
    gal = gallery or _default_gallery
    if gal is None:
        raise _lib.RevoError("search(): no gallery; create a Gallery and call set_default(gallery=...)")
    return gal.search(queries, k=k, score_threshold=score_threshold)


# ---- profiler ---------------------------------------------------------------
def prof_enable(on=True):
    """on: False/0 off, True/1 every kernel class, 2 only the body GEMMs, 3 every fourth launch of each
    body-GEMM class (cheap enough for a timed region)."""
    _lib.load().revo_prof_enable(int(on))


def prof_reset():
    _lib.load().revo_prof_reset()


def prof_report():
    buf = C.create_string_buffer(1 << 16)
    _lib.check(_lib.load().revo_prof_report(buf, len(buf)), "revo_prof_report")
    return json.loads(buf.value.decode())
