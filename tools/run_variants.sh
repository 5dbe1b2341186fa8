for d in "" "-DVPL_X_NOMFMA" "-DVPL_X_NOLOAD" "-DVPL_X_NOTICKET" "-DVPL_X_NOLOAD -DVPL_X_NOMFMA"; do
  echo "=== defs: $d"
  VPL_EXTRA_DEFS="$d" timeout -k 10 300 python tools/dbg_stamps_step.py 2>&1 | grep -A1 "window 256" | tail -1
done
