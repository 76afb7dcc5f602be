"""Multi-GPU plumbing: one process per GPU.

The exchange itself is native: libcobs_gpu.so calls RCCL (ncclAllGather / grouped
ncclSend + ncclRecv over xGMI) through the cobs_gpu_comm_* / cobs_gpu_batch_exchange_* entry
points of include/cobs_gpu_batch.h (cobs_amd/csrc/comm.cpp); `Comm` below binds them.
torch.distributed is only the LAUNCHER there: it hands rank 0's unique id to the other
ranks.  The torch-level functions further down (all_gather_counts, assemble_counts,
merge_hits) are the same exchange written against torch.distributed; they run on any
backend, which is what lets world_size-2 gloo tests cover the N > 1 logic on CPU and lets
several ranks share one GPU in tests (RCCL refuses two ranks on one device).

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
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _capi


class Comm:
    """One rank of a native RCCL communicator (cobs_gpu_comm, comm.cpp)."""

    def __init__(self, unique_id, rank, nranks, device=-1):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        assert len(unique_id) == _capi.UNIQUE_ID_BYTES
        buf = (C.c_uint8 * _capi.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _capi.check(self._lib.cobs_gpu_comm_create(buf, rank, nranks, device, C.byref(self._h)))

    @staticmethod
    def unique_id():
        """ncclGetUniqueId: call on one rank, hand the 128 bytes to all ranks"""
        lib = _capi.load()
        buf = (C.c_uint8 * _capi.UNIQUE_ID_BYTES)()
        _capi.check(lib.cobs_gpu_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch(cls, group=None, device=-1):
        """torch.distributed as the launcher: broadcast rank 0's unique id, then ncclCommInitRank"""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, device)

    @property
    def rank(self):
        return int(self._lib.cobs_gpu_comm_rank(self._h))

    @property
    def size(self):
        """ncclCommCount of the communicator"""
        return int(self._lib.cobs_gpu_comm_size(self._h))

    def set_timeout(self, timeout_ms):
        """the stream waits the library performs around collectives give up after this long (0 = never): the
        communicator is aborted and the call fails with ERR_RCCL instead of waiting for a peer that never arrives"""
        self._lib.cobs_gpu_comm_set_timeout(self._h, int(timeout_ms))

    def state(self):
        """one line: what this rank's communicator entered last -- callable from a watchdog thread while the owner
        thread sits inside a call"""
        if not getattr(self, "_h", None):
            return "communicator closed"
        buf = C.create_string_buffer(512)
        n = self._lib.cobs_gpu_comm_state(self._h, buf, len(buf))
        return buf.raw[:n].decode("utf-8", "replace")

    def preflight(self, timeout_ms=20000, big_bytes=0):
        """collective: the first bytes of a new communicator, each step under a time limit, every byte checked (uneven
        grouped send / receive all-to-all, all-gather, all-reduce; big_bytes > 0: a timed all-to-all of that size per
        pair).  -> dict of sizes and microseconds; raises CobsGpuError (the communicator is then unusable)"""
        out = (C.c_uint64 * 8)()
        _capi.check(self._lib.cobs_gpu_comm_preflight(self._h, int(timeout_ms), int(big_bytes), C.byref(out)))
        res = {"alltoall_bytes": int(out[0]), "alltoall_us": int(out[1]), "allgather_us": int(out[2]),
               "allreduce_us": int(out[3])}
        if out[4]:
            res["big_alltoall_bytes_received"] = int(out[4])
            res["big_alltoall_us"] = int(out[5])
            res["big_alltoall_recv_GBps"] = round(out[4] / max(out[5], 1) / 1e3, 2)
        return res

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cobs_gpu_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


def exchange_counts_by_plan(local, layouts, total_counts, nq, mode=_capi.XCHG_ALLTOALL, group=None):
    """The count exchange exactly as libcobs_gpu.so plans it (cobs_gpu_exchange_plan, comm.cpp: plan_exchange),
    executed with torch.distributed point-to-point transfers instead of grouped ncclSend / ncclRecv: every rank
    sends and receives the byte ranges the native plan names and applies its assembly copies.  Any backend --
    world_size > 1 gloo tests run the library's own plan across real processes (a size mismatch between a
    sender's and a receiver's plan would hang or fail here, as it would on xGMI).
    local: [nq, n_local] integer tensor (this rank's count rows); layouts: per rank, per file
    (slot_begin, slot_count, doc_offset).  -> (q_begin, q_count, rows [q_count, total_counts])."""
    lib = _capi.load()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nfiles = len(layouts[0])
    begins = (C.c_uint64 * (world * nfiles))(*[f[0] for lay in layouts for f in lay])
    counts = (C.c_uint64 * (world * nfiles))(*[f[1] for lay in layouts for f in lay])
    docoff = (C.c_uint64 * nfiles)(*[f[2] for f in layouts[0]])
    eb = local.element_size()
    xf = (_capi.Xfer * world)()
    cp = (_capi.Copy2D * (world * nfiles))()
    ncp = C.c_size_t(world * nfiles)
    out = (C.c_uint64 * 6)()
    _capi.check(lib.cobs_gpu_exchange_plan(begins, counts, docoff, world, nfiles, total_counts, nq, eb, mode, rank,
                                           xf, cp, C.byref(ncp), out))
    q0, qn, staging_bytes, global_bytes, use_ag, my_row = [int(v) for v in out]
    mine = local.contiguous().view(torch.uint8).reshape(-1)
    staging = torch.zeros(max(staging_bytes, 1), dtype=torch.uint8, device=local.device)
    if use_ag:      # one ncclAllGather: every rank's nq * row bytes, rank after rank
        parts = [torch.empty(nq * my_row, dtype=torch.uint8, device=local.device) for _ in range(world)]
        dist.all_gather(parts, mine[:nq * my_row].contiguous(), group=group)
        for j in range(world):
            staging[xf[j].recv_offset: xf[j].recv_offset + xf[j].recv_bytes] = parts[j]
    else:
        reqs, keep = [], []
        for j in range(world):
            if j == rank:
                continue
            if xf[j].send_bytes:
                t = mine[xf[j].send_offset: xf[j].send_offset + xf[j].send_bytes].contiguous()
                keep.append(t)
                reqs.append(dist.isend(t, dist.get_global_rank(group, j) if group is not None else j, group=group))
            if xf[j].recv_bytes:
                t = torch.empty(xf[j].recv_bytes, dtype=torch.uint8, device=local.device)
                keep.append((t, xf[j].recv_offset))
                reqs.append(dist.irecv(t, dist.get_global_rank(group, j) if group is not None else j, group=group))
        for r in reqs:
            r.wait()
        for k in keep:
            if isinstance(k, tuple):
                staging[k[1]: k[1] + k[0].numel()] = k[0]
    got = torch.zeros(max(global_bytes, 1), dtype=torch.uint8, device=local.device)
    for c in list(cp)[:ncp.value]:
        src = mine if c.src_is_local else staging
        for h in range(c.height):
            so, do = c.src_offset + h * c.src_pitch, c.dst_offset + h * c.dst_pitch
            got[do:do + c.width] = src[so:so + c.width]
    rows = got[:global_bytes].view(local.dtype).reshape(qn, total_counts) if qn else got[:0].view(local.dtype).reshape(0, total_counts)
    return q0, qn, rows


def exchange_hits_by_plan(local_hits, nq, group=None):
    """The owner-routed hit exchange as libcobs_gpu.so plans it (cobs_gpu_hit_exchange_plan), executed with
    torch.distributed: local_hits = this rank's (query, file, doc, score) records; rank j owns the queries
    [nq*j/N, nq*(j+1)/N).  -> (q_begin, q_count, the records of those queries from every rank)."""
    lib = _capi.load()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    owner = lambda q: max(j for j in range(world) if nq * j // world <= q)
    buckets = [[h for h in local_hits if owner(h[0]) == j] for j in range(world)]
    mine = [len(b) for b in buckets]
    allc = [None] * world
    dist.all_gather_object(allc, mine, group=group)
    flat = (C.c_uint64 * (world * world))(*[c for row in allc for c in row])
    xf = (_capi.Xfer * world)()
    out = (C.c_uint64 * 2)()
    _capi.check(lib.cobs_gpu_hit_exchange_plan(flat, world, rank, xf, out))
    flat_send = [v for b in buckets for h in b for v in h]
    send = torch.tensor(flat_send, dtype=torch.int32).reshape(len(flat_send) // 4, 4).view(torch.uint8).reshape(-1)
    peer = (lambda j: dist.get_global_rank(group, j)) if group is not None else (lambda j: j)
    assert send.numel() == int(out[1])
    recv = torch.zeros(max(int(out[0]), 1), dtype=torch.uint8)
    reqs, keep = [], []
    for j in range(world):
        if j == rank:
            recv[xf[j].recv_offset: xf[j].recv_offset + xf[j].recv_bytes] = send[xf[j].send_offset: xf[j].send_offset + xf[j].send_bytes]
            continue
        if xf[j].send_bytes:
            t = send[xf[j].send_offset: xf[j].send_offset + xf[j].send_bytes].contiguous()
            keep.append(t)
            reqs.append(dist.isend(t, peer(j), group=group))
        if xf[j].recv_bytes:
            t = torch.empty(xf[j].recv_bytes, dtype=torch.uint8)
            keep.append((t, xf[j].recv_offset))
            reqs.append(dist.irecv(t, peer(j), group=group))
    for r in reqs:
        r.wait()
    for k in keep:
        if isinstance(k, tuple):
            recv[k[1]: k[1] + k[0].numel()] = k[0]
    recs = recv[:int(out[0])].view(torch.int32).reshape(int(out[0]) // 16, 4).tolist()
    return nq * rank // world, nq * (rank + 1) // world - nq * rank // world, [tuple(r) for r in recs]


class ShardedSearch:
    """cobs_index.Search over an index sharded by sub-index block across the ranks
    of `group`.  Every rank calls the same methods with the same arguments and
    gets the same (global) results.

    transport "rccl": the native exchange of libcobs_gpu.so (one GPU per rank);
    "torch": the torch.distributed restatement (any backend; several ranks may share a GPU);
    "auto": rccl when the group's backend is nccl, else torch.
    hbm_budget > 0: every rank streams what of its shard does not fit (BASELINE configs[4])."""

    def __init__(self, path, group=None, device=-1, hbm_budget=0, transport="auto", shard_mode=0):
        from .search import Batch, Search
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if transport == "auto":
            transport = "rccl" if dist.get_backend(group) == "nccl" else "torch"
        self.transport = transport
        self.search_local = Search(path, device=device, shard_rank=self.rank, shard_count=self.world,
                                   hbm_budget=hbm_budget, shard_mode=shard_mode)
        self.batch = Batch(self.search_local)
        self.total_counts = self.search_local.total_counts
        self.comm = Comm.from_torch(group, device) if transport == "rccl" else None
        self.layouts = shard_slots(self.search_local, group) if transport == "torch" else None

    def counts(self, queries):
        """-> [Q, total_counts] tensor of per-document counts on every rank"""
        self.batch.set_queries(queries)
        self.batch.run(0.0)
        if self.comm is not None:
            self.batch.exchange_counts(self.comm, _capi.XCHG_ALLGATHER)
            self.batch.sync()
            return self.batch.global_counts_tensor()[2]
        self.batch.sync()
        # torch transport: the library's own exchange plan (what comm.cpp executes over RCCL), moved with
        # torch.distributed transfers
        _, _, rows = exchange_counts_by_plan(self.batch.counts_tensor(), self.layouts, self.total_counts, len(queries),
                                             _capi.XCHG_ALLGATHER, self.group)
        return rows

    def search_hits(self, queries, threshold=0.0, num_results=0):
        s = self.search_local
        if self.comm is not None:
            return s.sharded_search_hits(self.comm, queries, threshold, num_results)
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


__all__ = ["Comm", "ShardedSearch", "shard_slots", "all_gather_counts", "assemble_counts", "merge_hits",
           "exchange_counts_by_plan", "exchange_hits_by_plan"]
