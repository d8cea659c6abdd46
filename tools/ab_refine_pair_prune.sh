#!/bin/bash
# r06: does the pair pruning of k_refine cost the three-lane headline anything?  It added 5.5 M vector wave-instructions per 96-frame launch
# (35.8 -> 41.3 M) for the same duration alone on the chip; the three-lane step is bound by vector issue.
# A = the product library, B = the same sources with refine_pair's test compiled out (built by hand into line-mod-pipeline_amd/lib_alt/, see profiles/r06_ab_experiments.log section 6).
# Alternates A B A B A B; default lanes of configs 2 and 5.
set -u
OUT=${1:-gpurun_out/r06_refine_pair_prune}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$OUT"
LIB=$R/line-mod-pipeline_amd/lib/liblinemod_hip.so
cp "$LIB" /tmp/lib_A.so
cp "$R/line-mod-pipeline_amd/lib_alt/liblinemod_hip_nopairprune.so" /tmp/lib_B.so
for i in 1 2 3; do
  for v in A B; do
    cp /tmp/lib_$v.so "$LIB"
    for cfg in 2 5; do
      python3 "$R/bench.py" --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e > "$R/$OUT/c${cfg}_${v}_$i.json" 2> "$R/$OUT/c${cfg}_${v}_$i.err"
      python3 -c "
import json,sys
d=json.load(open('$R/$OUT/c${cfg}_${v}_$i.json'))
print('config $cfg lib $v run $i: %.1f det/s  %.4f ms/step  stages %s' % (d['value'], d['ms_per_step'], d['roofline']['stage_us_per_frame_one_lane']))"
    done
  done
done
cp /tmp/lib_A.so "$LIB"
