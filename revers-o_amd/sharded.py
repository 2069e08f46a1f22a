"""Row-sharded gallery search across the GPUs of one node.

One process per GPU; rank r owns gallery rows [offset_r, offset_r + n_r).  A
search is: (optionally) all-gather the data-parallel query blocks, every rank scans
its own shard for ALL queries with global row ids, one all-gather of the per-shard
top-k (k * 12 bytes per query and rank -- latency bound on xGMI, no all-reduce),
then the same merge on every rank.  The reference has no distributed code at all
(single process, core_system.py:650-664); this is the build's scale-out of that call.

The two compute steps are injected (``local_search`` / ``merge``) so that the
protocol -- offsets, gather layout, ordering -- can be exercised on CPU with the
gloo backend by the tests; the product wiring (:func:`from_gallery`) uses the HIP
kernels and nothing else.
"""
import torch
import torch.distributed as dist


class ShardedSearch:
    def __init__(self, local_search, merge, local_rows, group=None):
        """local_search(queries, k, threshold, index_offset) -> (scores[Q,k], idx[Q,k], counts[Q]);
        merge(part_scores[P,Q,k], part_idx[P,Q,k], k, threshold) -> same triple."""
        self.local_search = local_search
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_rows = int(local_rows)
        self.offset, self.total_rows = self._exchange_offsets()

    @classmethod
    def from_gallery(cls, gallery, group=None):
        from . import engine
        return cls(
            lambda q, k, thr, off: gallery.search(q, k, thr, index_offset=off),
            lambda ps, pi, k, thr: engine.merge_topk(ps, pi, k, thr),
            len(gallery), group)

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

    def refresh(self, local_rows):
        """Call after the local shard grew."""
        self.local_rows = int(local_rows)
        self.offset, self.total_rows = self._exchange_offsets()

    def _all_gather(self, out, inp):
        """all_gather_into_tensor; device tensors on a gloo group (a rehearsal of the N > 1 path on one GPU,
        or a CPU-only interconnect) are staged through host memory, RCCL takes them as they are."""
        if inp.is_cuda and dist.get_backend(self.group) != "nccl":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, inp.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)

    def gather_queries(self, local_queries):
        """Data-parallel embed leaves [B, D] on every rank; all ranks need all queries."""
        if self.world == 1:
            return local_queries
        out = torch.empty((self.world * local_queries.shape[0], local_queries.shape[1]),
                          dtype=local_queries.dtype, device=local_queries.device)
        self._all_gather(out, local_queries)
        return out

    def search(self, queries, k, threshold=None):
        """queries: identical [Q, D] on every rank.  Returns the global top-k triple on every rank."""
        s, i, c = self.local_search(queries, k, None, self.offset)   # threshold applies after the merge
        if self.world == 1:
            return self.merge(s[None], i[None], k, threshold)
        Q = queries.shape[0]
        # concatenated along dim 0 (the layout both RCCL and gloo accept), viewed as [world, Q, k]
        ps = torch.empty((self.world * Q, k), dtype=s.dtype, device=s.device)
        pi = torch.empty((self.world * Q, k), dtype=i.dtype, device=i.device)
        self._all_gather(ps, s)
        self._all_gather(pi, i)
        return self.merge(ps.view(self.world, Q, k), pi.view(self.world, Q, k), k, threshold)
