R=$GRAFT_REPO_ROOT
for f in 1 0 2 1 0; do python3 $R/bench.py --config 2 --scan-form $f --no-cpu-baseline --no-h2d --no-pose-e2e --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); l=d['latency']
print('form $f', l['resident_frame'], l['pinned_host_frame']['median_us'], l['pageable_host_frame']['median_us'], l['gpu_stage_us_resident'])"; done
