#!/bin/bash
# One GPU call: BA parity tests, bench, k_lin stamps (plus experiment variants given as arguments: "-DNAME" ...)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solve.py tests/test_config5_batch.py tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/t.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/b.json 2> gpurun_out/b.err; echo "bench rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/b.json'):
    if ln.startswith('{'):
        d=json.loads(ln); print('value',d['value'],'ms',d['ms_per_step'],'parity',d.get('parity'))
        agg={}
        for l in d['launches']: agg[l['kernel']]=agg.get(l['kernel'],0)+l['ms']
        print({k:round(v,4) for k,v in agg.items()})
        print([ (l['kernel'],round(l['ms'],4)) for l in d['launches'] if l['kernel'] in ('k_lin','k_cost')])
PY
timeout -k 10 300 python tools/dbg_stamps_lin.py > gpurun_out/stamps_lin.log 2>&1; grep -A2 "^window 0" gpurun_out/stamps_lin.log
for x in "$@"; do
  VPL_EXTRA_DEFS="${x//,/ }" timeout -k 10 300 python tools/dbg_stamps_lin.py > gpurun_out/stamps_lin_$x.log 2>&1; echo "variant $x"; grep -A1 "^window 0" gpurun_out/stamps_lin_$x.log
done
