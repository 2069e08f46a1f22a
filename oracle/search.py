"""CPU oracle: brute-force cosine similarity + top-k, the search half of the path.

TEST INFRASTRUCTURE ONLY (see oracle/pe_vit.py header).  PARITY UNPINNED: the
arithmetic lives in qdrant-client (``requirements.txt:39`` ``>=1.3.0``, local /
embedded mode), absent from ``/root/reference`` and this image; the reference
has no tests or golden vectors.  Restated from the reference's call sites:

* ``core_system.py:600-603``  collection created with ``Distance.COSINE``
* ``core_system.py:608``      vectors stored as plain float lists (float32 rows)
* ``core_system.py:657-664``  one query vector, ``limit=max_results``,
  ``score_threshold=similarity_threshold``
* ``core_system.py:666``      empty result list when nothing passes the threshold

and qdrant-client's local-mode behaviour as recalled in SURVEY.md §8(a) a8/a10:
gallery rows L2-normalised at insert, query normalised, ``scores = G @ q``,
descending order, stop at ``limit`` or at the first ``score < score_threshold``.

Tie order: upstream's ``argsort()[::-1]`` is not stable, so exact duplicates
(which real revers-o galleries contain, ``core_system.py:406-408``) come back in
arbitrary order there.  This oracle fixes the order as (score desc, index asc);
tests compare tie groups accordingly.
"""
import numpy as np


def normalize_rows(x, dtype=np.float32):
    """L2-normalise rows; a zero row stays zero (qdrant local guards the same way)."""
    x = np.asarray(x, dtype=dtype)
    n = np.linalg.norm(x.astype(np.float64), axis=-1, keepdims=True)
    n = np.where(n == 0.0, 1.0, n)
    return (x.astype(np.float64) / n).astype(dtype)


def cosine_scores(gallery, queries, acc=np.float64):
    """[N,D] x [Q,D] -> [Q,N] cosine of already-normalised rows, accumulated in
    ``acc`` and rounded once to float32 (the score dtype the path returns)."""
    g = np.asarray(gallery, dtype=acc)
    q = np.asarray(queries, dtype=acc)
    return (q @ g.T).astype(np.float32)


def rank_topk(scores_row, k, threshold=None):
    """Indices of the k best entries of one score row, (score desc, index asc),
    cut at ``score >= threshold``."""
    n = scores_row.shape[0]
    order = np.lexsort((np.arange(n), -scores_row.astype(np.float64)))
    order = order[:k]
    if threshold is not None:
        order = order[scores_row[order] >= np.float32(threshold)]
    return order


def search(gallery, queries, k, threshold=None, normalize=True, acc=np.float64):
    """Brute-force cosine top-k.

    Returns (scores float32 [Q,k], indices int64 [Q,k], counts int32 [Q]) padded
    with -inf / -1 past ``counts`` — the layout of the build's ``search()``.
    """
    g = normalize_rows(gallery) if normalize else np.asarray(gallery, dtype=np.float32)
    q = normalize_rows(queries) if normalize else np.asarray(queries, dtype=np.float32)
    Q = q.shape[0]
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    out_c = np.zeros((Q,), dtype=np.int32)
    if g.shape[0] == 0:
        return out_s, out_i, out_c
    # block over queries so a 1M-row gallery does not materialise Q x N at once
    step = max(1, min(Q, (1 << 27) // max(1, g.shape[0])))
    g_acc = g.astype(acc)
    for s in range(0, Q, step):
        sc = (q[s:s + step].astype(acc) @ g_acc.T).astype(np.float32)
        for r in range(sc.shape[0]):
            row = sc[r]
            kk = min(k, row.shape[0])
            # partial selection first, then exact ordering of the survivors
            if row.shape[0] > 4 * kk + 16:
                kth = np.partition(row, row.shape[0] - kk)[row.shape[0] - kk]
                cand = np.nonzero(row >= kth)[0]
            else:
                cand = np.arange(row.shape[0])
            o = cand[np.lexsort((cand, -row[cand].astype(np.float64)))][:kk]
            if threshold is not None:
                o = o[row[o] >= np.float32(threshold)]
            c = o.shape[0]
            out_s[s + r, :c] = row[o]
            out_i[s + r, :c] = o
            out_c[s + r] = c
    return out_s, out_i, out_c


def search_chunked(chunks, queries, k, threshold=None, normalize_queries=True, acc=np.float64):
    """:func:`search` over a gallery that arrives in row chunks -- ``chunks`` yields ``(start, rows float32 [n, D])`` with
    the rows ALREADY normalised (what a store hands back: qdrant normalises at insert, core_system.py:600-603) -- so that
    the BASELINE gallery (1 M x 1024: 4 GB of float32, 8 GB as float64) can be ranked without holding it twice.  Same
    arithmetic and order rule as :func:`search`: float64 accumulation, one rounding to float32, (score desc, index asc),
    cut at ``score >= threshold``.  The Q x N float32 score matrix IS materialised (keep Q small)."""
    q = normalize_rows(queries) if normalize_queries else np.asarray(queries, dtype=np.float32)
    qa = q.astype(acc)
    parts, starts = [], []
    for start, rows in chunks:
        parts.append((qa @ np.asarray(rows, dtype=np.float32).astype(acc).T).astype(np.float32))
        starts.append(int(start))
    Q = q.shape[0]
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    out_c = np.zeros((Q,), dtype=np.int32)
    if not parts:
        return out_s, out_i, out_c
    order = np.argsort(starts)
    assert all(starts[order[j]] + parts[order[j]].shape[1] == starts[order[j + 1]] for j in range(len(order) - 1)), \
        "chunks must tile the gallery"
    sc = np.concatenate([parts[j] for j in order], axis=1)
    for r in range(Q):
        o = rank_topk(sc[r], min(k, sc.shape[1]), threshold) if sc.shape[1] <= 4 * k + 16 else None
        if o is None:
            row = sc[r]
            kth = np.partition(row, row.shape[0] - k)[row.shape[0] - k]
            cand = np.nonzero(row >= kth)[0]
            o = cand[np.lexsort((cand, -row[cand].astype(np.float64)))][:k]
            if threshold is not None:
                o = o[row[o] >= np.float32(threshold)]
        c = o.shape[0]
        out_s[r, :c] = sc[r][o]
        out_i[r, :c] = o
        out_c[r] = c
    return out_s, out_i, out_c


def search_one_reference_style(gallery_f32_normalized, query, k, threshold=None):
    """The reference's actual usage: ONE query (``region_embeddings[0]``,
    core_system.py:657), float32 ``G @ q``, full argsort, walk until ``limit`` or
    the first score below the threshold.  Used for the cpu_baseline timing."""
    q = np.asarray(query, dtype=np.float32)
    n = np.linalg.norm(q)
    if n > 0:
        q = q / n
    scores = gallery_f32_normalized @ q
    order = np.argsort(scores)[::-1]
    out = []
    for idx in order:
        if len(out) >= k:
            break
        if threshold is not None and scores[idx] < threshold:
            break
        out.append((int(idx), float(scores[idx])))
    return out


def merge_topk(part_scores, part_indices, k, threshold=None):
    """Merge per-shard top-k lists: inputs [P,Q,k] (padded with -inf/-1, indices
    already global) -> ([Q,k], [Q,k], [Q]) under (score desc, index asc)."""
    P, Q, kk = part_scores.shape
    s = np.transpose(part_scores, (1, 0, 2)).reshape(Q, P * kk)
    i = np.transpose(part_indices, (1, 0, 2)).reshape(Q, P * kk)
    out_s = np.full((Q, k), -np.inf, dtype=np.float32)
    out_i = np.full((Q, k), -1, dtype=np.int64)
    out_c = np.zeros((Q,), dtype=np.int32)
    for r in range(Q):
        valid = i[r] >= 0
        sv, iv = s[r][valid], i[r][valid]
        o = np.lexsort((iv, -sv.astype(np.float64)))[:k]
        if threshold is not None:
            o = o[sv[o] >= np.float32(threshold)]
        c = o.shape[0]
        out_s[r, :c] = sv[o]
        out_i[r, :c] = iv[o]
        out_c[r] = c
    return out_s, out_i, out_c


# ---------------------------------------------------------------------------------------------
# Two-phase sharded search (the build's scale-out of core_system.py:659-664; include/revo.h
# "the same search in two phases").  A CPU restatement of what one rank does between the two
# all-gathers, used by the gloo protocol test (tests/test_host_logic.py) and as the checker of
# the HIP path's sharded results.  "Scan scores" here are the exact fp32 scores (the HIP scan
# uses bf16 inputs; the protocol is the same).
def f32_orderable(x):
    """Order-preserving map float32 -> uint32 (include/revo.h: published scan scores)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def packed_bytes(n_queries, k):
    return (n_queries * k * 12 + n_queries * 4 + 15) // 16 * 16


class OracleShardBackend:
    """The backend interface of revers-o_amd/sharded.py on numpy (rows already normalised).

    ``scan_noise`` > 0 imitates the HIP path's bf16 scan: candidates are selected on scores perturbed by a
    deterministic pseudo-noise of at most that size (a function of query and global row only), results are re-scored
    exactly, and every shard publishes its certificate bound (best noisy score of a row it did not re-score +
    ``scan_noise``), so that the protocol's certificate check and second, exact round run on CPU too."""

    def __init__(self, shard_rows, scan_noise=0.0, row_offset=0):
        import torch
        self._torch = torch
        self.shard = np.asarray(shard_rows, dtype=np.float32)
        self.scan_noise = float(scan_noise)
        self.row_offset = int(row_offset)
        self._cand = None
        self.exact_calls = 0

    def _noisy(self, q):
        """[Q, n] scan scores: exact scores + bounded deterministic noise."""
        sc = cosine_scores(self.shard, q)
        if self.scan_noise > 0.0:
            rows = (np.arange(self.shard.shape[0], dtype=np.int64) + self.row_offset)[None, :]
            qi = np.arange(q.shape[0], dtype=np.int64)[:, None]
            h = ((rows * 2654435761 + qi * 40503 + 12345) % 1000003).astype(np.float64) / 1000003.0
            sc = (sc.astype(np.float64) + (2.0 * h - 1.0) * self.scan_noise).astype(np.float32)
        return sc

    def ksel(self, k):
        return 32 if k <= 16 else 64

    def packed_bytes(self, n_queries, k):
        return packed_bytes(n_queries, k)

    def search(self, queries, k, threshold):
        t = self._torch
        s, i, c = search(self.shard, np.asarray(queries, dtype=np.float32), k, threshold, normalize=False)
        return t.from_numpy(s), t.from_numpy(i), t.from_numpy(c)

    def candidates(self, queries, k, top_m):
        q = np.asarray(queries, dtype=np.float32)
        ksel = self.ksel(k)
        if self.scan_noise > 0.0:
            noisy = self._noisy(q)
            Q, n = noisy.shape
            kk = min(ksel, n)
            s = np.full((Q, ksel), -np.inf, dtype=np.float32)
            i = np.full((Q, ksel), -1, dtype=np.int64)
            c = np.full((Q,), kk, dtype=np.int32)
            for r in range(Q):
                o = np.lexsort((np.arange(n), -noisy[r].astype(np.float64)))[:kk]
                s[r, :kk], i[r, :kk] = noisy[r][o], o
        else:
            s, i, c = search(self.shard, q, ksel, None, normalize=False)
        self._cand = (q, s, i, c)
        pub = np.zeros((q.shape[0], top_m), dtype=np.uint32)
        for r in range(q.shape[0]):
            n = min(int(c[r]), top_m)
            pub[r, :n] = f32_orderable(s[r, :n])
        return self._torch.from_numpy(pub.view(np.int32))

    def finish(self, n_queries, k, all_bounds, index_offset):
        q, s, i, c = self._cand
        ksel = s.shape[1]
        out = np.zeros((packed_bytes(n_queries, k),), dtype=np.uint8)
        idx = np.full((n_queries, k), -1, dtype=np.int64)
        sc = np.full((n_queries, k), -np.inf, dtype=np.float32)
        cert = np.full((n_queries,), -np.inf, dtype=np.float32)
        for r in range(n_queries):
            bound = np.uint32(0)
            if all_bounds is not None:
                pub = np.sort(all_bounds[:, r, :].numpy().view(np.uint32).reshape(-1))[::-1]
                brank = min(64, 2 * ksel)                      # the rule of topk_finish_kernel (revers-o_amd/csrc/topk.hip)
                if pub.shape[0] >= brank:
                    bound = pub[brank - 1]
            nc = int(c[r])
            keep = [j for j in range(nc) if f32_orderable(s[r, j]) >= bound]
            # certificate bound: the first candidate the shard bound dropped, else the worst kept one of a full list
            if len(keep) < nc:
                cert[r] = s[r, len(keep)] + np.float32(self.scan_noise)
            elif nc == ksel and self.shard.shape[0] > ksel:
                cert[r] = s[r, ksel - 1] + np.float32(self.scan_noise)
            rows = i[r, keep]
            exact = cosine_scores(self.shard[rows], q[r:r + 1])[0] if len(keep) else np.zeros((0,), np.float32)
            o = np.lexsort((rows, -exact.astype(np.float64)))[:k]
            idx[r, :len(o)] = rows[o] + index_offset
            sc[r, :len(o)] = exact[o]
        self._pack(out, n_queries, k, idx, sc, cert)
        return self._torch.from_numpy(out)

    @staticmethod
    def _pack(out, n, k, idx, sc, cert):
        out[: n * k * 8] = idx.view(np.uint8).reshape(-1)
        out[n * k * 8: n * k * 12] = sc.view(np.uint8).reshape(-1)
        out[n * k * 12: n * k * 12 + n * 4] = cert.view(np.uint8).reshape(-1)

    def exact(self, q_idx, need, k, index_offset):
        """Second round: the exact local top-k of the listed queries (brute force on this shard)."""
        self.exact_calls += 1
        q = self._cand[0][np.asarray(q_idx, dtype=np.int64)]
        n = q.shape[0]
        s, i, c = search(self.shard, q, k, None, normalize=False)
        i = np.where(i >= 0, i + index_offset, i)
        out = np.zeros((packed_bytes(n, k),), dtype=np.uint8)
        self._pack(out, n, k, i.astype(np.int64), s, np.full((n,), -np.inf, dtype=np.float32))
        return self._torch.from_numpy(out)

    def merge(self, packed_all, parts, n_queries, k, threshold, certify=False):
        pb = packed_bytes(n_queries, k)
        buf = packed_all.numpy().reshape(parts, pb)
        pi = np.stack([buf[p, : n_queries * k * 8].copy().view(np.int64).reshape(n_queries, k) for p in range(parts)])
        ps = np.stack([buf[p, n_queries * k * 8: n_queries * k * 12].copy().view(np.float32).reshape(n_queries, k)
                       for p in range(parts)])
        s, i, c = merge_topk(ps, pi, k, threshold)
        t = self._torch
        if not certify:
            return t.from_numpy(s), t.from_numpy(i), t.from_numpy(c)
        cert = np.stack([buf[p, n_queries * k * 12: n_queries * k * 12 + n_queries * 4].copy().view(np.float32)
                         for p in range(parts)]).max(0)
        full = merge_topk(ps, pi, k, None)                       # the k-th merged score before the threshold cut
        sk = np.where(full[2] >= k, full[0][:, k - 1], -np.inf).astype(np.float32)
        need = np.maximum(sk, np.float32(threshold)) if threshold is not None else sk
        unc = np.nonzero(~((cert == -np.inf) | (need > cert)))[0].astype(np.int32)
        uq = np.zeros((max(n_queries, 1),), dtype=np.int32)
        un = np.zeros((max(n_queries, 1),), dtype=np.float32)
        uq[: unc.shape[0]] = unc[::-1]                           # "in no particular order": the caller must sort
        un[: unc.shape[0]] = need[unc[::-1]]
        return (t.from_numpy(s), t.from_numpy(i), t.from_numpy(c),
                (t.tensor([unc.shape[0]], dtype=t.int32), t.from_numpy(uq), t.from_numpy(un)))
