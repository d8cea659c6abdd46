R=$GRAFT_REPO_ROOT
for q in 4 8 12; do
  GPU_MAX_HW_QUEUES=$q python3 $R/bench.py --no-cpu-baseline --config 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['config']['h2d_inclusive']; print('queues $q config 2 ->', round(d['value']), 'det/s; h2d', h['value'], h['h2d_GBps'], 'GB/s')"
done
GPU_MAX_HW_QUEUES=8 python3 $R/bench.py --no-cpu-baseline --config 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['config']['h2d_inclusive']; print('queues 8 config 3 ->', round(d['value']), 'det/s; h2d', h['value'], h['h2d_GBps'], 'GB/s')"
