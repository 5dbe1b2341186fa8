#!/bin/bash
# Instruction-fetch counters of the BA kernels (own rocprofv3 passes, kernel trace only): is a kernel waiting for instructions?
# usage (GPU box, repo root): bash tools/pmc_icache.sh <out_dir>
set -e
ROOT=$(pwd); OUT=$ROOT/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/ic -o ic -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --output-format csv -d $OUT/wt -o wt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import sys, pandas as pd
out = sys.argv[1]
for f in ("ic/ic_counter_collection.csv", "wt/wt_counter_collection.csv"):
    d = pd.read_csv(out + "/" + f)
    d["k"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("vpl::", "")
    d = d[d["k"].str.startswith("k_")]
    print(d.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="max").round(0).to_string())
PY
