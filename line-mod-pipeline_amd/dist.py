"""Template-bank sharding across the GPUs of one node.

One process per GPU.  Every rank holds the same frames and the templates [n*r/R, n*(r+1)/R) of every
class (contiguous template_id ranges, global ids preserved -- lm_config.shard_rank/shard_size).  The
only exchange step of the path is the gather of the per-shard sorted match lists: one all-gather of
the counts and one all-gather of the packed records per BATCH of frames (torch.distributed: backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), followed by the R-way merge +
adjacent-unique on every rank (lm_merge_batch, host side of the C ABI).  The payload is tiny (20 B per
match, a few hundred KB per 128-frame step), so the collectives are latency-bound; batching the
frames of a step is what keeps them off the critical path.

The local matcher is injected (`local_match(threshold, class_idx) -> (records[B, cap], counts[B])`):
the product passes Detector.match_batch; the gloo tests pass the CPU oracle.
"""
import numpy as np

MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                        ("class_idx", "<i4")])


def shard_range(n, rank, size):
    """Template-id range of shard `rank` (same rule as lmh::shard_range in csrc/lm_host.h)."""
    return n * rank // size, n * (rank + 1) // size


def frame_range(n_frames, rank, size):
    """Frame-shard (bench.py --parallelism frame-shard): every rank holds the WHOLE bank and answers the
    frames [n*r/R, n*(r+1)/R) of a batch -- the same contiguous rule as the template shards.  Nothing is exchanged in the data
    path; a caller that wants all lists on one rank gathers them (gather_frame_lists)."""
    return n_frames * rank // size, n_frames * (rank + 1) // size


def gather_frame_lists(lists, n_frames, group=None):
    """The frame-shard counterpart of ShardGather: every rank hands in the lists of the frames it owns (frame_range order) and
    receives the lists of all n_frames frames, in frame order -- one all_gather_object (control plane of a test / a collector
    process; the bench's frame-shard mode exchanges nothing)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return list(lists)
    parts = [None] * world
    dist.all_gather_object(parts, [np.asarray(l) for l in lists], group=group)
    out = []
    for r in range(world):
        f0, f1 = frame_range(n_frames, r, world)
        if len(parts[r]) != f1 - f0:
            raise ValueError("rank %d delivered %d lists for its %d frames" % (r, len(parts[r]), f1 - f0))
        out.extend(parts[r])
    return out


class ShardGather:
    """All-gather + merge of per-shard match lists for a batch of frames.

    What travels is small and variable (about 20 B x matches), so the exchange is two collectives per
    BATCH: the per-frame counts of every rank ([R, B] int32), then the packed records padded to the
    largest rank total of this batch (rounded up so the buffers are reused).  The merge of all frames is
    one call (`merge_batch_fn`, lm_merge_batch: threaded R-way merge + unique); with only the per-frame
    `merge_fn` it falls back to a Python loop.  There is no per-frame capacity: the second collective is sized to the
    largest rank total of the batch, so whatever the local matcher returns is exchanged (the reference consumes ALL
    matches, HighLevelLinemod.cpp:206-253); `cap` is kept for callers that size their own buffers with it."""

    GRANULE = 4096      # records; gather buffers grow in these steps

    def __init__(self, merge_fn, group=None, cap=4096, device=None, pack_fn=None, merge_batch_fn=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.merge_fn, self.pack_fn, self.merge_batch_fn = merge_fn, pack_fn, merge_batch_fn
        self.group = group
        self.cap = cap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self._cnt = {}
        self._rec = {}

    def _count_buffers(self, B):
        if B not in self._cnt:
            t = self.torch
            self._cnt[B] = (t.zeros(B, dtype=t.int32, device=self.device),
                            t.zeros((self.world, B), dtype=t.int32, device=self.device))
        return self._cnt[B]

    def _record_buffers(self, n):
        n = max((n + self.GRANULE - 1) // self.GRANULE, 1) * self.GRANULE
        if n not in self._rec:
            t = self.torch
            self._rec[n] = (t.zeros(n * 5, dtype=t.int32, device=self.device),
                            t.zeros((self.world, n * 5), dtype=t.int32, device=self.device))
        return (n,) + self._rec[n]

    def _pack(self, records, counts):
        if self.pack_fn is not None:
            return self.pack_fn(records, counts)
        parts = [records[i, :int(counts[i])] for i in range(len(counts))]
        return np.concatenate(parts) if parts else np.zeros(0, MATCH_DTYPE)

    def gather_merge(self, records, counts):
        """records: structured [B, >=max(counts)] MATCH_DTYPE; counts: [B].  Returns a list of B merged
        match arrays, identical on every rank."""
        B = len(counts)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        if int(counts.max(initial=0)) > records.shape[1]:
            raise ValueError("a count exceeds the record array's stride")
        if self.world == 1:
            return [records[i, :counts[i]].copy() for i in range(B)]
        return self.gather_merge_packed(self._pack(records, counts), counts)

    def owned_frames(self, B):
        """Frames of a B-frame batch whose merge this rank does (contiguous ranges, like the template shards)."""
        return B * self.rank // self.world, B * (self.rank + 1) // self.world

    def gather_merge_packed(self, packed, counts, owned_only=False):
        """The same for lists that are already packed back to back (lm_pack_matches layout).  owned_only: every
        rank still receives all lists, but merges only the frames it owns (owned_frames) and returns
        (first_frame, lists): the host work of the exchange then does not grow with the number of ranks."""
        B = len(counts)
        counts = np.array(counts, dtype=np.int32)            # writable copy (torch.from_numpy)
        if int(counts.sum()) != len(packed):
            raise ValueError("counts do not add up to the packed records")
        if self.world == 1:
            ends = np.cumsum(counts)
            lists = [packed[e - c:e].copy() for e, c in zip(ends, counts)]
            return (0, lists) if owned_only else lists
        f0, f1 = self.owned_frames(B) if owned_only else (0, B)
        t = self.torch
        # 1. counts of every rank
        csend, crecv = self._count_buffers(B)
        csend.copy_(t.from_numpy(counts))
        self.dist.all_gather_into_tensor(crecv.view(-1), csend, group=self.group)
        allc = crecv.cpu().numpy()                                  # [R, B]
        totals = allc.sum(axis=1)
        # 2. packed records, padded to the largest rank total
        stride, rsend, rrecv = self._record_buffers(int(totals.max()))
        if len(packed):
            rsend[:len(packed) * 5].copy_(t.from_numpy(np.ascontiguousarray(packed).view(np.int32).reshape(-1).copy()))
        self.dist.all_gather_into_tensor(rrecv.view(-1), rsend, group=self.group)
        allr = rrecv.cpu().numpy().view(MATCH_DTYPE).reshape(self.world, stride)
        # 3. merge every frame
        if self.merge_batch_fn is not None:
            merged, mc = self.merge_batch_fn(allr, allc, f0, f1) if owned_only else self.merge_batch_fn(allr, allc)
            ends = np.cumsum(mc)
            out = [merged[e - c:e] for e, c in zip(ends, mc)]
        else:
            starts = np.cumsum(allc, axis=1) - allc
            out = []
            for i in range(f0, f1):
                out.append(self.merge_fn([allr[r, starts[r, i]:starts[r, i] + allc[r, i]] for r in range(self.world)]))
        return (f0, out) if owned_only else out


class ShardedDetector:
    """Drop-in for Detector.match_batch when the bank is sharded over the ranks of a process group."""

    def __init__(self, local_match, merge_fn, group=None, cap=4096, device=None, pack_fn=None, merge_batch_fn=None):
        self.local_match = local_match
        self.gather = ShardGather(merge_fn, group=group, cap=cap, device=device, pack_fn=pack_fn,
                                  merge_batch_fn=merge_batch_fn)

    def match_batch(self, n_frames, threshold, class_idx=-1):
        records, counts = self.local_match(n_frames, threshold, class_idx)
        return self.gather.gather_merge(records, counts)
