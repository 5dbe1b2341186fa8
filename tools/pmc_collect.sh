#!/bin/bash
# Collects the HBM traffic counters of bench.py's kernels the way MI355X_MICROARCH.md prescribes: one rocprofv3 run per
# counter (kernel trace only, no other trace domain), plus the calibration program, then tools/pmc_summary.py.
# usage (on the GPU box, from the repo root): bash tools/pmc_collect.sh <out_dir> <out_json>
set -e
ROOT=$(pwd)
OUT=$ROOT/$1
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $OUT/calib tools/pmc_calib/calib.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cf -o cf -- $OUT/calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cw -o cw -- $OUT/calib > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/f/f_counter_collection.csv $OUT/w/w_counter_collection.csv $OUT/cf/cf_counter_collection.csv $OUT/cw/cw_counter_collection.csv $ROOT/$2 > $OUT/summary.txt
cp $OUT/f/f_counter_collection.csv $OUT/fetch.csv; cp $OUT/w/w_counter_collection.csv $OUT/write.csv
