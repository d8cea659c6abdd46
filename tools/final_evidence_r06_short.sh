#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > $O/r06_final_gpu_suite.log 2>&1; grep -E "passed|failed|error" $O/r06_final_gpu_suite.log | tail -2
bash tools/collect_counters.sh r06 "2 3 5" > $O/r06_collect.log 2>&1
cd $R
cp $O/r06_counters_c2.json $O/r06_counters_c3.json $O/r06_counters_c5.json profiles/
for c in 2 3 5; do
  python3 bench.py --config $c > $O/r06_bench_config$c.json 2> $O/r06_bench_config$c.err
  python3 -c "
import json
d=json.loads(open('$O/r06_bench_config$c.json').read().splitlines()[-1]); r=d['roofline']
print('config $c', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['bound'], r['frac'], 'launch', r['avg_launch_us'], 'counter', r.get('counter_file'), 'pipeline', (d.get('roofline_pipeline') or {}).get('frac'), 'lat', ((d.get('latency') or {}).get('resident_frame') or {}).get('median_us'), 'e2e', (d.get('pose_e2e') or {}).get('us_per_frame'), (d.get('pose_e2e') or {}).get('us_per_frame_pipelined_pinned_frames'), (d.get('pose_e2e') or {}).get('us_per_frame_serial'))"
done
: > $O/r06_fuzz_final.log
for s in 31 32; do timeout 140 python3 tests/fuzz_parity.py 100 $s 2>&1 | tail -1 >> $O/r06_fuzz_final.log; done
cat $O/r06_fuzz_final.log
