# scratch: one bench line per config (value, one-lane stage times, loads kept); extra args are passed on
R=$GRAFT_REPO_ROOT
for c in 2 3; do python3 $R/bench.py --no-cpu-baseline --no-h2d --config $c "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config $c $* ->', round(d['value']), 'det/s', d['roofline']['stage_us_per_frame_one_lane'], d['roofline']['pruning']['feature_loads_kept'])"; done
