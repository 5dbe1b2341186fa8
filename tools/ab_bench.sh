#!/bin/bash
# A/B of two kernel variants inside ONE gpurun call (same box, same clocks): builds with the given -D switches, runs
# bench.py twice per variant.  usage: bash tools/ab_bench.sh "<defs A>" "<defs B>"
for V in "$1" "$2"; do
  VPL_EXTRA_DEFS="$V" python -c "
import sys; sys.path.insert(0,'.')
import vplines_slam_amd._build as b; b.build_hip(verbose=False, force=True)"
  for k in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernels_ms_per_step']; print('[$V]', round(d['value']), ' '.join('%s %.4f' % (n, t) for n, t in sorted(k.items(), key=lambda x: -x[1])[:9]))"
  done
done
