for d in "-DVPL_MARG_NOISE_REL=0.0" "-DVPL_MARG_NOISE_REL=1e-13" "-DVPL_MARG_NOISE_REL=1e-9"; do
  echo "=== $d"
  VPL_EXTRA_DEFS="$d" python -c "
import importlib.util,os
spec=importlib.util.spec_from_file_location('b','vplines-slam_amd/_build.py'); b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build_hip(force=True)"
  timeout -k 10 300 python tools/dbg_chain.py 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' | head -14
done
