"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, xGMI between the GPUs of a node).

The reference has no distributed code; a compact index is a concatenation of
sub-indexes over disjoint, contiguous document ranges (reference
cobs/query/compact_index/mmap_search_file.cpp:22-27, search_file.cpp:30-32), so
the path shards by sub-index block: rank r stages and scans only its block and
produces the counts of its own documents.  One exchange per batch brings the
disjoint slices together:

  * counts mode (threshold 0, every document is a result): all-gather of the
    u8/u16/u32 count slices (as bytes -- RCCL has no 16-bit integer type);
  * hits mode (threshold > 0): each rank selects its hits on the device; the
    small (file, doc, score) lists are gathered and merged by
    (score desc, file asc, doc asc), the reference's result order
    (classic_search.cpp:136-145,179-188).

All functions work on whatever device the tensors live on, so the exchange and
merge logic is covered by world_size-2 gloo tests on CPU.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_slots(search, group=None):
    """-> list over ranks of per-file (slot_begin, slot_count, doc_offset) tuples"""
    mine = []
    for f in range(search.num_files):
        i = search.info(f)
        mine.append((int(i.slot_begin), int(i.slot_count), int(i.doc_offset)))
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, mine, group=group)
    return out


def all_gather_counts(local, group=None):
    """local: [Q, n_local] integer tensor (any width) on this rank; shards may
    differ in n_local.  -> list over ranks of [Q, n_r] tensors (same dtype)."""
    world = dist.get_world_size(group)
    q = local.shape[0]
    n = torch.tensor([local.shape[1]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes) if sizes else 0
    esz = local.element_size()
    # pad to the largest shard and move bytes (RCCL/NCCL have no 16-bit integer type)
    padded = torch.zeros((q, nmax), dtype=local.dtype, device=local.device)
    padded[:, :local.shape[1]] = local
    send = padded.view(torch.uint8).reshape(-1)
    recv = torch.empty((world, send.numel()), dtype=torch.uint8, device=local.device)
    try:
        dist.all_gather_into_tensor(recv.reshape(-1), send, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.stack(parts)
    out = []
    for r in range(world):
        t = recv[r].view(local.dtype).reshape(q, nmax)
        out.append(t[:, :sizes[r]])
    assert esz == out[0].element_size()
    return out


def assemble_counts(gathered, layouts, total_counts):
    """Scatter the per-rank slices into global document order.
    gathered[r]: [Q, n_r]; layouts[r]: per-file (slot_begin, slot_count, doc_offset);
    local rows hold the files' held slots back to back."""
    q = gathered[0].shape[0]
    full = torch.zeros((q, total_counts), dtype=gathered[0].dtype, device=gathered[0].device)
    for r, lay in enumerate(layouts):
        off = 0
        for (begin, count, doc_off) in lay:
            full[:, doc_off + begin: doc_off + begin + count] = gathered[r][:, off:off + count]
            off += count
    return full


def merge_hits(per_rank_hits, num_results=0, total_hashes=2, total_documents=None):
    """per_rank_hits: list over ranks of [(file, doc, score), ...] of ONE query.
    Reference order: score desc, then (file, doc) asc; document order when the
    query has a single hash in total (max_counts <= 1)."""
    allh = [h for hits in per_rank_hits for h in hits]
    if total_hashes > 1:
        allh.sort(key=lambda h: (-h[2], h[0], h[1]))
    else:
        allh.sort(key=lambda h: (h[0], h[1]))
    if total_documents is not None and num_results:
        num_results = min(num_results, total_documents)
    return allh[:num_results] if num_results else allh


class ShardedSearch:
    """cobs_index.Search over an index sharded by sub-index block across the ranks
    of `group`.  Every rank calls the same methods with the same arguments and
    gets the same (global) results."""

    def __init__(self, path, group=None, device=-1):
        from .search import Batch, Search
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.search_local = Search(path, device=device, shard_rank=self.rank, shard_count=self.world)
        self.batch = Batch(self.search_local)
        self.layouts = shard_slots(self.search_local, group)
        self.total_counts = self.search_local.total_counts

    def counts(self, queries):
        """-> [Q, total_counts] tensor of per-document counts on every rank"""
        self.batch.set_queries(queries)
        self.batch.run(0.0)
        self.batch.sync()
        gathered = all_gather_counts(self.batch.counts_tensor(), self.group)
        return assemble_counts(gathered, self.layouts, self.total_counts)

    def search_hits(self, queries, threshold=0.0, num_results=0):
        s = self.search_local
        self.batch.set_queries(queries)
        if num_results > 0:
            self.batch.run_topk(threshold, num_results)      # K3: only the k best of the shard leave the device
        else:
            self.batch.run(threshold)
        self.batch.sync()
        # local ranked hits of the held documents; truncation to num_results per shard is safe
        # because the global top-k is a subset of the union of the per-shard top-k
        local = [self.batch.hits_host(i, num_results) for i in range(len(queries))]
        everyone = [None] * self.world
        dist.all_gather_object(everyone, local, group=self.group)
        out = []
        for i, q in enumerate(queries):
            th = sum((len(q) - s.info(f).term_size + 1) * s.info(f).num_hashes for f in range(s.num_files))
            out.append(merge_hits([everyone[r][i] for r in range(self.world)], num_results, th,
                                  self.total_counts))
        return out

    def search(self, query, threshold=0.0, num_results=0):
        from .search import SearchResult
        hits = self.search_hits([query], threshold, num_results)[0]
        return [SearchResult(self.search_local.doc_name(f, d), sc) for (f, d, sc) in hits]


__all__ = ["ShardedSearch", "shard_slots", "all_gather_counts", "assemble_counts", "merge_hits"]
