#!/bin/bash
# round 4, GPU call AR: does a run with the shipped MIOpen tables still search (and append) configurations?  One default bench (640 px / 16) and one
# configs[4] bench (1280 px / 8), then the per-user table directory against the shipped files.
set -o pipefail
O=gpurun_out/r4ar; mkdir -p $O/miopen
export TAMTR_MIOPEN_DB_DIR=$PWD/$O/userdb
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -E "timed|capture" $O/bench.err | cut -c1-200
timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 3 --imgsz 1280 --batch 8 > $O/bench1280.json 2> $O/bench1280.err; echo "bench 1280 rc=$?"; grep -E "timed|capture" $O/bench1280.err | cut -c1-200
for f in $O/userdb/*/*.txt; do
  b=$(basename $f); echo "$b: user copy $(wc -l < $f) lines / $(stat -c %s $f) bytes, shipped $(wc -l < tam-tr_amd/tuned/miopen/$b) lines / $(stat -c %s tam-tr_amd/tuned/miopen/$b) bytes"
  cp $f $O/miopen/$b
done
