R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05_final
cd $R
for c in 2 3 5; do python3 bench.py --config $c > gpurun_out/r05_final/bench_config$c.json 2> gpurun_out/r05_final/bench_config$c.err; tail -c 300 gpurun_out/r05_final/bench_config$c.json | head -c 10 > /dev/null; python3 -c "
import json
d=json.loads(open('gpurun_out/r05_final/bench_config$c.json').read().splitlines()[-1]); r=d['roofline']
print('config $c', d['value'], 'ms/step', d['ms_per_step'], 'roofline', r['bound'], r['frac'], r['kernel'], 'pipeline', (d.get('roofline_pipeline') or {}).get('frac'), 'h2d', (d['config'].get('h2d_inclusive') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'lat', ((d.get('latency') or {}).get('resident_frame') or {}).get('median_us'), 'e2e', (d.get('pose_e2e') or {}).get('us_per_frame'), (d.get('pose_e2e') or {}).get('us_per_frame_pipelined_pinned_frames'))"; done
python3 bench.py > gpurun_out/r05_final/bench_default.json 2> gpurun_out/r05_final/bench_default.err; tail -c 200 gpurun_out/r05_final/bench_default.json
