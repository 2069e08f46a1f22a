"""Row-sharded gallery search across the GPUs of one node.

One process per GPU; rank r owns gallery rows [offset_r, offset_r + n_r).  The reference has no
distributed code at all (single process, one collection, core_system.py:650-664); this is the build's
scale-out of that call.  A search of Q queries (identical on every rank) for the best k:

  1. every rank scans its own shard for ALL queries (bf16 MFMA scan, the same kernels as the single-GPU
     search) and keeps its best ``ksel`` candidates per query; it publishes the scan scores of the best
     ``top_m`` of them                                                        [Q, top_m] int32
  2. all-gather #1 of those scores (4 * top_m bytes per query and rank)        [P, Q, top_m]
     -> the j-th largest, j = min(64, 2 ksel), is a lower bound of the j-th best scan score over the whole gallery
  3. every rank re-scores in fp32 only its candidates at or above that bound (about j / P per query
     instead of ksel: the fp32 row gathers, the part of a search that does not shrink with the shard,
     shrink with it; j is twice what the unsharded search re-scores, so that the certificate of step 5
     all but never fails and step 6 stays the exception) and takes its local top k with global row ids
  4. all-gather #2 of the packed per-rank results (12 * k bytes per query and rank, one buffer)
  5. the same merge on every rank (score descending, global row id ascending), which also checks every query's
     exactness certificate over all shards (include/revo.h, "EXACTNESS": each shard ships, next to its results, the
     best fp32 score a row it did NOT re-score can have)
  6. only if some query fails that check (none on ordinary data; near-duplicate clusters wider than the scan's
     candidate lists): every rank re-does those queries exactly on its shard (collecting pass + fp32 re-score, brute
     force if need be) and a second packed all-gather + merge replaces their results.  The list of such queries is the
     same on every rank (same merged data), so the ranks take this branch together.

When the shards scan against an ESTIMATE of the whole gallery's admission level (the default on the HIP backend once the
shard sizes are known: revo_search_set_total_rows), steps 1-3's exchange is left out: each
shard's list is already cut at that level, and the certificate of step 5 counts the estimate as the score an unseen row
may have.  One all-gather per search, then.

Both exchanges are latency-bound on xGMI (kilobytes to a few MB, no all-reduce).  The result equals the
unsharded search of the concatenated gallery bit for bit: both are the top-k of an exhaustive fp32 scoring.

The compute steps come from a *backend* object so that the protocol -- offsets, gather layouts, ordering --
can be exercised on CPU with the gloo backend by the tests; the product wiring (:func:`from_gallery`) uses
the HIP kernels and nothing else.  Backend interface:

    ksel(k) -> int
    estimates(k) -> bool                                              (optional, with set_total_rows: shards scan against an estimated level)
    search(queries, k, threshold) -> (scores, indices, counts)        the whole search on one shard (world size 1)
    candidates(queries, k, top_m) -> int32 [Q, top_m]
    finish(n_queries, k, all_bounds [P, Q, top_m], index_offset) -> uint8 [packed_bytes(Q, k)]
    packed_bytes(n_queries, k) -> int
    merge(packed_all uint8 [P * packed_bytes], parts, n_queries, k, threshold, certify=False)
        -> (scores, indices, counts[, (unc_count int32 [1], unc_q int32 [Q], unc_need fp32 [Q])])
    exact(q_idx int32 [n], need fp32 [n], k, index_offset) -> uint8 [packed_bytes(n, k)]
"""
import torch
import torch.distributed as dist


class GalleryBackend:
    """The product backend: a device-resident :class:`engine.Gallery` shard."""

    def __init__(self, gallery):
        from . import engine
        self._engine = engine
        self.gallery = gallery
        self._buf = {}          # packed result blocks, reused from search to search (no allocation in the steady state)

    def _packed(self, tag, n_queries, k):
        key = (tag, int(n_queries), int(k))
        if key not in self._buf:
            if len(self._buf) > 16:
                self._buf.clear()
            self._buf[key] = torch.empty((self.packed_bytes(n_queries, k),), dtype=torch.uint8, device=self.gallery.device)
        return self._buf[key]

    def ksel(self, k):
        return self._engine.search_ksel(k)

    def estimates(self, k):
        return self._engine.search_estimates(k)

    def packed_bytes(self, n_queries, k):
        return self._engine.packed_bytes(n_queries, k)

    def search(self, queries, k, threshold):
        return self.gallery.search(queries, k, threshold)

    def candidates(self, queries, k, top_m):
        return self.gallery.search_candidates(queries, k, top_m)

    def finish(self, n_queries, k, all_bounds, index_offset):
        out = self._packed("finish", n_queries, k)
        # the threshold is applied after the merge (a per-element predicate: same result, one code path)
        self.gallery.search_finish(n_queries, k, all_bounds, None, index_offset, out_packed=out)
        return out

    def merge(self, packed_all, parts, n_queries, k, threshold, certify=False):
        return self._engine.merge_topk_packed(packed_all, parts, n_queries, k, threshold, certify=certify)

    def set_total_rows(self, total_rows):
        self.gallery.set_total_rows(total_rows)

    def exact(self, q_idx, need, k, index_offset):
        out = self._packed("exact", int(q_idx.shape[0]), k)
        self.gallery.search_exact(q_idx, need, k, index_offset, out_packed=out)
        return out


class ShardedSearch:
    def __init__(self, backend, local_rows, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_rows = int(local_rows)
        self.last_uncertified = 0          # queries of the last finished search that needed the second round
        self.second_rounds = 0             # searches that ran the protocol's second round on the shard handle's own candidates
        self.redone_searches = 0           # pipelined searches whose uncertified queries had to be searched again (PendingSearch)
        self._estimating, self._unc_rate = False, 0.0      # see _note_second_rounds
        self._est_off_at = 0               # total rows when the estimate was switched off for this gallery (0: never)
        self._timing = None                # enable_timing(): {tag: [event pairs or seconds]}
        self._timed_searches = 0
        self._gen = 0                      # searches started (whose candidates the shard handle holds: PendingSearch)
        self._pinned = None                # host landing places of the uncertified counts (search_async)
        self._gbuf = {}                    # gather targets, reused from search to search
        self.offset, self.total_rows = self._exchange_offsets()
        self._tell_total()
        # eight landing places (pinned where there is a device; plain host words on the CPU tier, which runs the same
        # slot bookkeeping so that the gloo tests cover it)
        self._pinned = [torch.zeros((1,), dtype=torch.int32) for _ in range(8)]
        if torch.cuda.is_available():
            self._pinned = [t.pin_memory() for t in self._pinned]
        self._inflight = [None] * 8

    def _gather_buf(self, tag, shape, dtype, device):
        key = (tag, tuple(shape), dtype)
        if key not in self._gbuf:
            if len(self._gbuf) > 16:
                self._gbuf.clear()
            self._gbuf[key] = torch.empty(shape, dtype=dtype, device=device)
        return self._gbuf[key]

    @classmethod
    def from_gallery(cls, gallery, group=None):
        return cls(GalleryBackend(gallery), len(gallery), group)

    def _device(self):
        if dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def _exchange_offsets(self):
        if self.world == 1:
            return 0, self.local_rows
        mine = torch.tensor([self.local_rows], dtype=torch.int64, device=self._device())
        allrows = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allrows, mine, group=self.group)
        sizes = [int(t.item()) for t in allrows]
        return sum(sizes[: self.rank]), sum(sizes)

    def _note_second_rounds(self, n, n_queries):
        """The shards start their scans from an ESTIMATE of the whole gallery's admission level (_tell_total), tuned to
        the tail that unit vectors in many dimensions give.  On data with a lighter tail it comes out too high, the
        merge's certificate then fails for many queries and every search pays the second round -- exact, but slower
        than not estimating at all.  So the share of uncertified queries is watched (a running mean; the count is the
        same on every rank, so every rank decides alike) and above 2 % the estimate is switched off for this gallery."""
        if not self._estimating or n_queries <= 0:
            return
        self._unc_rate = 0.8 * self._unc_rate + 0.2 * (n / n_queries)
        if self._unc_rate > 0.02:
            tell = getattr(self.backend, "set_total_rows", None)
            if tell is not None:
                tell(0)
            self._estimating = False
            self._est_off_at = max(1, self.total_rows)

    def _tell_total(self):
        """The shard's scans start from an estimate of the WHOLE gallery's admission level (backends that can use it).
        An estimate that was switched off for this gallery's data (_note_second_rounds) stays off until the gallery has
        doubled: more rows of the same kind do not change its tail."""
        tell = getattr(self.backend, "set_total_rows", None)
        if tell is None or self.world <= 1:
            return
        if self._est_off_at and self.total_rows < 2 * self._est_off_at:
            tell(0)
            self._estimating = False
            return
        tell(self.total_rows)
        self._estimating, self._unc_rate, self._est_off_at = True, 0.0, 0

    def refresh(self, local_rows):
        """Call after the local shard grew."""
        self.local_rows = int(local_rows)
        self.offset, self.total_rows = self._exchange_offsets()
        self._tell_total()

    # -- measured exchange times (bench.py: `allgather_ms`, `exchanges_per_search`) ------------------------------
    def enable_timing(self, on=True):
        """Time every exchange of the searches that follow: HIP events on the current stream around each all-gather of
        device tensors (RCCL enqueues there), wall clock for host tensors (the CPU tier's gloo tests)."""
        self._timing = {} if on else None
        self._timed_searches = 0

    def timing_report(self):
        """{"allgather_ms": {tag: mean ms per exchange}, "exchanges_per_search": mean count, "searches": n}; tags:
        "queries" (gather_queries), "bounds" (exchange 1, skipped when the shards estimate), "packed" (exchange 2),
        "packed_second_round" (exchange 3, rare)."""
        if self._timing is None:
            return None
        ms, n_ex = {}, 0
        for tag, recs in self._timing.items():
            vals = []
            for r in recs:
                if isinstance(r, tuple):
                    r[1].synchronize()
                    vals.append(r[0].elapsed_time(r[1]))
                else:
                    vals.append(r * 1e3)
            if vals:
                ms[tag] = sum(vals) / len(vals)
            if tag != "queries":
                n_ex += len(vals)
        n = max(self._timed_searches, 1)
        return {"allgather_ms": ms, "exchanges_per_search": n_ex / n, "searches": self._timed_searches}

    def _all_gather(self, out, inp, tag=None):
        """all_gather_into_tensor; device tensors on a gloo group (a rehearsal of the N > 1 path on one GPU,
        or a CPU-only interconnect) are staged through host memory, RCCL takes them as they are."""
        timed = self._timing is not None and tag is not None and self.world > 1
        if timed:
            if inp.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                import time
                t0 = time.perf_counter()
        if self.world == 1:
            out.copy_(inp.reshape(out.shape))
        elif inp.is_cuda and dist.get_backend(self.group) != "nccl":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, inp.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)
        if timed:
            if inp.is_cuda:
                e1.record()
                self._timing.setdefault(tag, []).append((e0, e1))
            else:
                self._timing.setdefault(tag, []).append(time.perf_counter() - t0)

    def gather_queries(self, local_queries):
        """Data-parallel embed leaves [B, D] on every rank; all ranks need all queries."""
        if self.world == 1:
            return local_queries
        out = torch.empty((self.world * local_queries.shape[0], local_queries.shape[1]),
                          dtype=local_queries.dtype, device=local_queries.device)
        self._all_gather(out, local_queries, "queries")
        return out

    def top_m(self, k):
        """Scores published per query and rank.  The finish step re-scores what reaches the min(64, 2 ksel)-th largest
        of all published scores: world * top_m >= that many makes it the exact global order statistic whenever no
        shard holds more than top_m of those rows; at least 8 so that an uneven spread of the best rows over the
        shards rarely loosens it."""
        ksel = self.backend.ksel(k)
        return min(ksel, max(8, -(-min(64, 2 * ksel) // self.world)))

    def search(self, queries, k, threshold=None):
        """queries: identical [Q, D] on every rank.  Returns the global top-k triple on every rank."""
        return self.search_async(queries, k, threshold).result()

    def search_async(self, queries, k, threshold=None):
        """The same search with its one host decision deferred: everything of the first round (scan, both exchanges,
        merge with the cross-shard certificate) is enqueued and a :class:`PendingSearch` comes back at once;
        ``result()`` waits for the count of uncertified queries and, only if there are any, runs the second round.
        A caller with a stream of query batches enqueues batch i + 1 before asking for batch i's result: the device never
        waits for the host between searches (``search()`` = ``search_async().result()`` does, once per call).
        Every rank must issue the same calls in the same order (the collectives are matched by order).
        ``queries`` must stay unmodified until ``result()`` has returned: a pending search whose shard state has been
        overwritten by a later one re-searches its uncertified rows from this tensor."""
        Q = queries.shape[0]
        top_m = self.top_m(k)
        if self.world == 1:
            return PendingSearch(self, self.backend.search(queries, k, threshold), None, None, queries, k, threshold, 0)
        if self._timing is not None:
            self._timed_searches += 1
        # The landing place this search will use may still belong to a search eight back that was never asked for its
        # result: finish that one NOW, before the shard handle's candidates become this search's.  Its result() may run
        # a repair search of its own (which advances _gen and takes the handle), so the slot is looked up again
        # afterwards -- flushed behind candidates(), as this did before, the repair overwrote this search's candidates
        # and the PendingSearch below was filed under the repair's generation.
        while self._inflight[(self._gen + 1) % len(self._pinned)] is not None:
            self._inflight[(self._gen + 1) % len(self._pinned)].result()
        self._gen += 1                                                           # the shard handle's candidates are this search's now
        gen = self._gen                                                          # (one value for the slot AND the PendingSearch)
        mine = self.backend.candidates(queries, k, top_m)                        # [Q, top_m] int32
        estimated = bool(self._estimating and self.backend.estimates(k))
        if estimated:
            # the shard scanned against an estimate of the whole gallery's admission level and its finish step cuts its list
            # there: that is what exchange 1 would have told it, so the exchange is left out (measured: the fp32 re-score is
            # no dearer without it -- 0.075 against 0.081 ms at 10 000 queries on a 125 k-row shard)
            packed = self.backend.finish(Q, k, None, self.offset)
        else:
            allb = self._gather_buf("bounds", (self.world * Q, top_m), mine.dtype, mine.device)
            self._all_gather(allb, mine, "bounds")                               # exchange 1: admission scores
            packed = self.backend.finish(Q, k, allb.view(self.world, Q, top_m), self.offset)
        allp = self._gather_buf("packed", (self.world * packed.numel(),), torch.uint8, packed.device)
        self._all_gather(allp, packed, "packed")                                 # exchange 2: packed per-rank top-k
        scores, idx, counts, unc = self.backend.merge(allp, self.world, Q, k, threshold, certify=True)
        # the count goes to (pinned) host memory behind the merge; nobody waits for it here
        slot = gen % len(self._pinned)               # (free: see the flush above)
        host_n, ev = self._pinned[slot], None
        if unc[0].is_cuda:
            host_n.copy_(unc[0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host_n.copy_(unc[0].reshape(1))          # CPU backends (tests)
        p = PendingSearch(self, (scores, idx, counts), unc, (host_n, ev), queries, k, threshold, gen)
        p._estimated = estimated
        self._inflight[slot] = p
        return p

    def _second_round(self, res, unc, n, k, threshold):
        """The uncertified queries (identical on every rank: same merged data), in one canonical order, exactly on every shard."""
        scores, idx, counts = res
        qs, order = torch.sort(unc[1][:n])
        need = unc[2][:n][order].contiguous()
        packed2 = self.backend.exact(qs.contiguous(), need, k, self.offset)
        allp2 = torch.empty((self.world * packed2.numel(),), dtype=torch.uint8, device=packed2.device)
        self._all_gather(allp2, packed2, "packed_second_round")                  # exchange 3 (rare)
        s2, i2, c2 = self.backend.merge(allp2, self.world, n, k, threshold)
        rows = qs.long()
        scores[rows] = s2
        idx[rows] = i2
        counts[rows] = c2
        return scores, idx, counts


class PendingSearch:
    """A sharded search whose first round is enqueued (ShardedSearch.search_async)."""

    def __init__(self, owner, res, unc, host, queries, k, threshold, gen):
        self._owner, self._res, self._unc, self._host = owner, res, unc, host
        self._queries, self._k, self._thr, self._gen = queries, k, threshold, gen
        self._note = True
        self._estimated = False                     # this search's scans started from the estimated admission level
        self._done = unc is None

    def result(self):
        """(scores, indices, counts), final: certified exact, or re-done exactly by the second round."""
        if self._done:
            return self._res
        o = self._owner
        host_n, ev = self._host
        if ev is not None:
            ev.synchronize()
        n = int(host_n[0])
        # give the landing place back BEFORE anything below starts another search: with eight searches pending the
        # repair search of this one would find "itself" in its slot and ask it for its result again
        slot = self._gen % len(o._inflight)
        if o._inflight[slot] is self:
            o._inflight[slot] = None
        o.last_uncertified = n
        if self._note and self._estimated:          # (a limit-50 search's second rounds say nothing about the estimate)
            o._note_second_rounds(n, int(self._queries.shape[0]))
        if n > 0:
            if o._gen == self._gen:
                o.second_rounds += 1
                self._res = o._second_round(self._res, self._unc, n, self._k, self._thr)
            else:
                # Another search has used the shard handle since (its candidates are gone).  Only the n uncertified queries
                # are searched again -- a search of their own, in step on every rank (the list is the same everywhere),
                # whose own second round finds its candidates in place.  Exact results do not depend on the batch a query
                # is searched in, so the patched rows are the rows a second round would have given.  (Searching the whole
                # batch again, as this did before, also made every search enqueued behind it stale: a cascade.)
                o.redone_searches += 1
                scores, idx, counts = self._res
                rows = torch.sort(self._unc[1][:n])[0].long()
                # the repair batch is padded to a power of two (its last row repeated): the gather and result buffers are
                # cached by shape, and one shape per uncertified count would push the steady-state shapes out of the caches
                nb = 1 << max(0, int(n) - 1).bit_length()
                pad = torch.cat([rows, rows[-1:].expand(nb - n)]) if nb > n else rows
                sub = o.search_async(self._queries[pad].contiguous(), self._k, self._thr)
                sub._note = False                       # the rate of uncertified queries is counted once per query
                s2, i2, c2 = sub.result()
                scores[rows] = s2[:n]
                idx[rows] = i2[:n]
                counts[rows] = c2[:n]
                o.last_uncertified = n
                self._res = (scores, idx, counts)
        self._done = True
        return self._res


class LocalShards:
    """The same protocol over several shards held by ONE process (a list of backends, e.g. the eight shards of a
    1 M-row gallery on one GPU in the tests and in scripts/sharded_stage_bench.py; or one handle per GPU of a
    single-process deployment): the "all-gathers" are concatenations.  ``offsets[p]`` = global id of shard p's row 0."""

    def __init__(self, backends, offsets, estimating=False):
        self.backends = list(backends)
        self.offsets = [int(o) for o in offsets]
        self.last_uncertified = 0
        self.estimating = bool(estimating)       # the shards scan against the whole gallery's estimated level: no bound exchange
        self._galleries = None                   # from_galleries: the handles whose total-row setting this object owns
        self._told = 0
        self._unc_rate = 0.0

    @classmethod
    def from_galleries(cls, galleries):
        """Shards held as engine.Gallery handles.  While this object lives the handles estimate the whole gallery's
        admission level (set_total_rows); close() -- or dropping the object -- puts them back to searching on their own."""
        self = cls([GalleryBackend(g) for g in galleries], [0] * len(galleries), estimating=len(galleries) > 1)
        self._galleries = list(galleries)
        self._retell()
        return self

    def _retell(self):
        """Offsets and the total follow the shards' current sizes (a shard may have grown since the last search)."""
        if self._galleries is None:
            return
        offs, tot = [], 0
        for g in self._galleries:
            offs.append(tot)
            tot += len(g)
        self.offsets = offs
        if self.estimating and tot != self._told:
            for g in self._galleries:
                g.set_total_rows(tot)
            self._told = tot

    def _note_second_rounds(self, n, n_queries):
        """ShardedSearch._note_second_rounds' rule: above 2 % uncertified queries (running mean) the estimate is off."""
        if not self.estimating or n_queries <= 0:
            return
        self._unc_rate = 0.8 * self._unc_rate + 0.2 * (n / n_queries)
        if self._unc_rate > 0.02:
            self.estimating = False
            for b in self.backends:
                tell = getattr(b, "set_total_rows", None)
                if tell is not None:
                    tell(0)
            self._told = 0

    def close(self):
        """Give the handles back: they no longer belong to a sharded gallery."""
        if self._galleries is not None:
            for g in self._galleries:
                try:
                    g.set_total_rows(0)
                except Exception:
                    pass
            self._galleries, self._told = None, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, queries, k, threshold=None):
        self._retell()
        P, Q = len(self.backends), queries.shape[0]
        ksel = self.backends[0].ksel(k)
        top_m = min(ksel, max(8, -(-min(64, 2 * ksel) // P)))
        allb = torch.stack([b.candidates(queries, k, top_m) for b in self.backends])            # [P, Q, top_m]
        estimated = bool(self.estimating and self.backends[0].estimates(k))
        if estimated:
            allb = None                                                                          # (see ShardedSearch.search_async)
        allp = torch.cat([b.finish(Q, k, allb, off) for b, off in zip(self.backends, self.offsets)])
        scores, idx, counts, unc = self.backends[0].merge(allp, P, Q, k, threshold, certify=True)
        n = int(unc[0].item())
        self.last_uncertified = n
        if estimated:
            self._note_second_rounds(n, Q)
        if n > 0:
            qs, order = torch.sort(unc[1][:n])
            need = unc[2][:n][order].contiguous()
            allp2 = torch.cat([b.exact(qs.contiguous(), need, k, off) for b, off in zip(self.backends, self.offsets)])
            s2, i2, c2 = self.backends[0].merge(allp2, P, n, k, threshold)
            rows = qs.long()
            scores[rows] = s2
            idx[rows] = i2
            counts[rows] = c2
        return scores, idx, counts
