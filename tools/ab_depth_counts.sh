# r06 A/B on one box: config 5 end to end (bench.py's pose_e2e leg) with the depth checks' early verdicts from GPU counts (default) and from the host's crop pass
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for mode in gpu host; do
  if [ $mode = host ]; then export LM_E2E_HOST_DEPTH_COUNTS=1; else unset LM_E2E_HOST_DEPTH_COUNTS; fi
  python3 $R/bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-latency 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pose_e2e']
print('depth counts on the $mode:', 'pageable', p['us_per_frame'], 'pinned', p['us_per_frame_pipelined_pinned_frames'], 'serial', p['us_per_frame_serial'], '| post wall', p['pipelined']['post_us_per_frame'], '| depth cpu us/frame', p['pipelined']['post_cpu_us_per_frame_by_part']['depth_check'], '| hot path', d['value'])"
done; done
