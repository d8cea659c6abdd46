#!/bin/bash
# A/B: batches begun ahead of the one being collected (1 = two in flight, 2 = three) on the e2e child left by bench.py (LM_POSE_E2E_KEEP)
LM_POSE_E2E_KEEP=/tmp/e2e python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-latency > /dev/null 2>&1
cd /tmp/e2e
for a in 1 2 1 2 1 2; do LM_E2E_AHEAD=$a ./pose_e2e_bench bench.bank poses.bin frames.raw 1280 960 8 80 30 0 16 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ahead $a', d['poses_identical_across_passes'], [(k, round(d[k]['us_per_frame']), round(d[k]['in_begin_us_per_frame']), round(d[k]['waiting_for_the_gpu_us_per_frame']), round(d[k]['post_us_per_frame'])) for k in ('serial','pipelined','pipelined_pinned')])"; done
