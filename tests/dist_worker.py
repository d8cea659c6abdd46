"""Worker of tests/test_dist.py: one rank of a world_size-R gloo group on CPU.

Exercises the product's shard-gather layer (line-mod-pipeline_amd/dist.py) exactly as bench.py uses it
on RCCL, with the CPU oracle standing in for the per-shard GPU matcher (allowed: tests only)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lm = importlib.import_module("line-mod-pipeline_amd")
    synth = importlib.import_module("line-mod-pipeline_amd.synth")
    distmod = importlib.import_module("line-mod-pipeline_amd.dist")
    from oracle import oracle as O

    B, thr, cap = 3, 70.0, 2048
    frames = [synth.make_frame(640, 480, seed=300 + i) for i in range(B)]
    o = O.Detector(color_only=False)
    o.prepare(*frames[0])
    q = {(l, m): o.stage(0, l, m).reshape(480 >> l, 640 >> l) for l in range(2) for m in range(2)}
    descs, feats, _ = synth.make_bank(90, 2, 2, seed=17, quantized=q, crop_fraction=0.3)
    o.add_class("c", descs, feats)
    n = o.class_num_templates(0)
    lo, hi = distmod.shard_range(n, rank, world)

    def local_match(n_frames, threshold, class_idx):
        rec = np.zeros((n_frames, cap), lm.MATCH_DTYPE)
        cnt = np.zeros(n_frames, np.int32)
        for i in range(n_frames):
            m = o.match(frames[i][0], frames[i][1], threshold, class_idx, tid_lo=lo, tid_hi=hi)
            rec[i, :len(m)] = m
            cnt[i] = len(m)
        return rec, cnt

    # the product's configuration: packing and the merge of the whole batch in C (lm_pack_matches / lm_merge_batch)
    sd = distmod.ShardedDetector(local_match, lm.merge_matches, cap=cap, pack_fn=lm.pack_matches,
                                 merge_batch_fn=lm.merge_batch)
    merged = sd.match_batch(B, thr, 0)
    # frame-partitioned merge (what bench.py uses): this rank merges only the frames it owns
    rec, cnt = local_match(B, thr, 0)
    f0, owned = sd.gather.gather_merge_packed(lm.pack_matches(rec, cnt), cnt, owned_only=True)
    ok_owned = (f0, f0 + len(owned)) == sd.gather.owned_frames(B)
    for k, lst in enumerate(owned):
        ok_owned &= lst.tobytes() == merged[f0 + k].tobytes()
    # and the pure-Python fallback of the same exchange
    sd2 = distmod.ShardedDetector(local_match, lm.merge_matches, cap=cap)
    merged2 = sd2.match_batch(B, thr, 0)
    ok = ok_owned
    for i in range(B):
        full = o.match(frames[i][0], frames[i][1], thr, 0)
        ok &= len(full) > 0 and merged[i].tobytes() == full.tobytes() and merged2[i].tobytes() == full.tobytes()
    # a list longer than any fixed capacity (threshold 0: every template matches everywhere it was a candidate, far more
    # than the 4096 records a device-sorted list holds) is exchanged like any other: the sharded path never refuses what
    # the unsharded one returns (the reference consumes ALL matches, HighLevelLinemod.cpp:206-253)
    big = 1 << 16
    m0 = o.match(frames[0][0], frames[0][1], 0.0, 0, tid_lo=lo, tid_hi=hi)
    full0 = o.match(frames[0][0], frames[0][1], 0.0, 0)
    ok &= len(full0) > 4096 and len(m0) <= big
    rec0 = np.zeros((1, big), lm.MATCH_DTYPE)
    rec0[0, :len(m0)] = m0
    small = distmod.ShardGather(lm.merge_matches, cap=1, pack_fn=lm.pack_matches, merge_batch_fn=lm.merge_batch)
    got0 = small.gather_merge(rec0, np.array([len(m0)], np.int32))
    ok &= got0[0].tobytes() == full0.tobytes()
    # frame-shard: every rank holds the WHOLE bank and answers only its own frames of a batch; the union of the ranks'
    # lists, in frame order, is what one process returns for the whole batch -- and nothing but that gather crosses the ranks
    NF = 5
    fr = [synth.make_frame(640, 480, seed=300 + i) for i in range(NF)]
    f0, f1 = distmod.frame_range(NF, rank, world)
    mine = [o.match(fr[i][0], fr[i][1], thr, 0) for i in range(f0, f1)]
    every = distmod.gather_frame_lists(mine, NF)
    ok &= len(every) == NF and sum(len(l) for l in every) > 0
    covered = sorted(sum([list(range(*distmod.frame_range(NF, r, world))) for r in range(world)], []))
    ok &= covered == list(range(NF))                                        # every frame has exactly one owner
    if rank == 0:
        for i in range(NF):
            ok &= every[i].tobytes() == o.match(fr[i][0], fr[i][1], thr, 0).tobytes()
    dist.barrier()
    dist.destroy_process_group()
    print("RANK %d %s" % (rank, "OK" if ok else "FAIL"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
