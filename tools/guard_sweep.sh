#!/bin/bash
# every randomised sweep once more with VPL_DEBUG_GUARDS=1: the 64 bytes behind each device array hold 0xA5 and close() of a
# context raises if a kernel wrote there (GPU box).  Output: one line per sweep.
export VPL_DEBUG_GUARDS=1
out=${1:-gpurun_out/guards}
mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python "$@" > $out/$name.txt 2>&1; rc=$?; echo "$name rc=$rc guard lines: $(grep -c -i 'guard' $out/$name.txt) | $(tail -n 1 $out/$name.txt | cut -c1-160)"; }
run sequence tools/fuzz_sequence.py 150 6 21
run parity tools/fuzz_parity.py 10 8 21
run map tools/fuzz_map.py 12 6 21
run frontend tools/fuzz_frontend.py 40 21
run frontend2 tools/fuzz_frontend2.py 40 21
run frontend3 tools/fuzz_frontend3.py 30 21
