#!/bin/bash
# Collects the HBM traffic counters of bench.py's kernels the way MI355X_MICROARCH.md prescribes: one rocprofv3 run per
# counter (kernel trace only, no other trace domain), plus the calibration program, then tools/pmc_summary.py.
# usage (on the GPU box, from the repo root): bash tools/pmc_collect.sh <out_dir> <out_json> [<flops_json>]
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
# FP64 instruction counters (second roofline of bench.py): their own passes, kernel trace only
if [ -n "$3" ]; then
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/fl1 -o fl1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/fl2 -o fl2 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $ROOT
  python3 tools/pmc_flops_summary.py $ROOT/$3 $OUT/fl1/fl1_counter_collection.csv $OUT/fl2/fl2_counter_collection.csv > $OUT/flops_summary.txt
fi
