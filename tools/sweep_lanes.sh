# Lane / batch shape sweep of the default bench (frames resident): which shape is fastest after a round's kernel changes.
# --batch is frames per STEP (all lanes together).
R=$GRAFT_REPO_ROOT
for cfg in "2 3 288" "2 2 256" "2 3 384" "2 4 384" "2 4 288" "2 3 192" "2 2 192" "3 2 256" "3 3 288" "3 3 384" "3 2 384" "3 2 192" "5 2 16" "5 3 24" "5 4 32" "5 1 8"; do
  set -- $cfg
  python3 $R/bench.py --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e --steps 100 --warmup 10 --config $1 --lanes $2 --batch $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config $1 lanes $2 batch $3 ->', round(d['value']), 'det/s', d['ms_per_step'])"
done
