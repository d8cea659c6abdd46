# A/B of the scan kernels (LM_TUNE_SCAN_FORM): the nibble scan k_scan4 (1) against the bit-plane scan k_scan1 (2), one lane and the default lanes
R=$GRAFT_REPO_ROOT
for c in ${CONFIGS:-2 5 3}; do
  for f in ${FORMS:-1 2}; do
    python3 $R/bench.py --config $c --scan-form $f ${EXTRA} --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e --steps ${STEPS:-60} --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('config $c form $f $EXTRA ->', round(d['value']), 'det/s | one lane', r['stage_us_per_frame_one_lane'], '| scan launch', r['avg_launch_us'], 'us | kept', r['pruning']['feature_loads_kept'], r['pruning']['lane_loads_kept'], '| L1', r.get('scan1_lanes_per_frame'), 'survivors/frame', r.get('scan1_survivors_per_frame'), 'cand/frame', d['config']['list_lengths']['scan_candidates_per_frame'])"
  done
done
