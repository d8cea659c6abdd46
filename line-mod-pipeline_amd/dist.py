"""Template-bank sharding across the GPUs of one node.

One process per GPU.  Every rank holds the same frames and the templates [n*r/R, n*(r+1)/R) of every
class (contiguous template_id ranges, global ids preserved -- lm_config.shard_rank/shard_size).  The
only exchange step of the path is the gather of the per-shard sorted match lists: one all-gather of
the counts and one all-gather of fixed-capacity record buffers per BATCH of frames
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests),
followed by the R-way merge + adjacent-unique on every rank (lm_merge_matches, host side of the C
ABI).  The payload is tiny (20 B per match), so the collective is latency-bound; batching the
frames of a step into one collective is what keeps it off the critical path.

The local matcher is injected (`local_match(threshold, class_idx) -> (records[B, cap], counts[B])`):
the product passes Detector.match_batch; the gloo tests pass the CPU oracle.
"""
import numpy as np

MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("template_id", "<i4"),
                        ("class_idx", "<i4")])


def shard_range(n, rank, size):
    """Template-id range of shard `rank` (same rule as lmh::shard_range in csrc/lm_host.h)."""
    return n * rank // size, n * (rank + 1) // size


class ShardGather:
    """All-gather + merge of per-shard match lists for a batch of frames."""

    def __init__(self, merge_fn, group=None, cap=4096, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.merge_fn = merge_fn
        self.group = group
        self.cap = cap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self._bufs = {}

    def _buffers(self, B):
        if B not in self._bufs:
            t = self.torch
            # int32 view of the 20-byte records: [B, cap, 5]; counts ride in the same message
            send = t.zeros((B, self.cap * 5 + 1), dtype=t.int32, device=self.device)
            recv = t.zeros((self.world, B, self.cap * 5 + 1), dtype=t.int32, device=self.device)
            host = t.zeros((B, self.cap * 5 + 1), dtype=t.int32).pin_memory() if self.device.type == "cuda" else None
            self._bufs[B] = (send, recv, host)
        return self._bufs[B]

    def gather_merge(self, records, counts):
        """records: structured [B, >=max(counts)] MATCH_DTYPE; counts: [B].  Returns a list of B merged
        match arrays, identical on every rank."""
        B = len(counts)
        if int(counts.max(initial=0)) > self.cap:
            raise OverflowError("shard produced %d matches for one frame, gather capacity %d "
                                "(SURVEY.md 8e: K must cover all matches)" % (int(counts.max()), self.cap))
        if self.world == 1:
            return [records[i, :counts[i]].copy() for i in range(B)]
        t = self.torch
        send, recv, host = self._buffers(B)
        stage = np.zeros((B, self.cap * 5 + 1), np.int32)
        for i in range(B):
            n = int(counts[i])
            stage[i, :n * 5] = records[i, :n].view(np.int32).reshape(-1)
            stage[i, -1] = n
        if host is not None:
            host.copy_(t.from_numpy(stage))
            send.copy_(host, non_blocking=True)
        else:
            send.copy_(t.from_numpy(stage))
        self.dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        allr = recv.cpu().numpy()
        out = []
        for i in range(B):
            lists = []
            for r in range(self.world):
                n = int(allr[r, i, -1])
                lists.append(allr[r, i, :n * 5].copy().view(MATCH_DTYPE))
            out.append(self.merge_fn(lists))
        return out


class ShardedDetector:
    """Drop-in for Detector.match_batch when the bank is sharded over the ranks of a process group."""

    def __init__(self, local_match, merge_fn, group=None, cap=4096, device=None):
        self.local_match = local_match
        self.gather = ShardGather(merge_fn, group=group, cap=cap, device=device)

    def match_batch(self, n_frames, threshold, class_idx=-1):
        records, counts = self.local_match(n_frames, threshold, class_idx)
        return self.gather.gather_merge(records, counts)
