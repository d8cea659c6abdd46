R=$GRAFT_REPO_ROOT
for cfg in "3 2 256" "3 3 288" "3 3 384" "3 2 384" "3 2 192" "3 4 512" "3 2 256"; do
  set -- $cfg
  python3 $R/bench.py --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e --steps 60 --warmup 10 --config $1 --lanes $2 --batch $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config $1 lanes $2 batch $3 ->', round(d['value']), 'det/s', d['ms_per_step'], d['roofline']['kernel'])"
done
