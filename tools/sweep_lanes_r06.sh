# r06: lane / batch shapes of config 2 with the LDS-resident bit-plane scan (k_scanl likes launches whose workgroups fill whole rounds of the 256 CUs)
R=$GRAFT_REPO_ROOT
for cfg in "2 3 288" "2 3 384" "2 2 256" "2 4 384" "2 4 512" "2 3 192" "2 2 384" "2 4 256" "2 6 384" "2 3 576"; do
  set -- $cfg
  python3 $R/bench.py --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e --steps 60 --warmup 10 --config $1 --lanes $2 --batch $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config $1 lanes $2 batch $3 ->', round(d['value']), 'det/s', d['ms_per_step'], d['roofline']['stage_us_per_frame_one_lane'])"
done
