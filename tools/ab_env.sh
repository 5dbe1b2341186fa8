#!/bin/bash
# A/B of an environment switch of the library on one box: bash tools/ab_env.sh VAR "v1 v2 v1 v2" [bench args]
VAR=$1; VALS=$2; shift 2
for g in $VALS; do
  env $VAR=$g python bench.py --steps 40 --no-cpu-baseline --no-extras "$@" 2>/dev/null > /tmp/ab_$$.json
  python - "$VAR=$g" /tmp/ab_$$.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "solves/s", round(d["ms_per_step"], 4), "ms/step", {k: round(x, 3) for k, x in d["kernels_ms_per_step"].items()})
PY
done
