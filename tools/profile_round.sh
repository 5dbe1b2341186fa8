#!/bin/bash
# One profiling pass for a round (on the GPU box, from the repo root): rocprofv3 kernel stats of bench.py (BA) and of the
# front-end bench, then the PMC traffic passes.  usage: bash tools/profile_round.sh <round dir under gpurun_out> 
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba -o ba -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/ba.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python3 $ROOT/tools/bench_frontend.py > $OUT/fe.log 2>&1
cd $ROOT
bash tools/pmc_collect.sh gpurun_out/$1/pmc gpurun_out/$1/pmc_traffic.json gpurun_out/$1/pmc_flops.json
ls $OUT $OUT/ba $OUT/fe
# steady-state / long-track shapes (k_schur_mixed, k_marg<512>, k_schur<5>): kernel stats of tools/bench_tracks.py
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tracks -o tracks -- python3 $ROOT/tools/bench_tracks.py 512 > $OUT/tracks.log 2>&1
cd $ROOT
