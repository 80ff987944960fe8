#!/bin/bash
# Regenerate tam-tr_amd/tuned/miopen/* : one bench run in MIOpen's timed-search mode writing into gpurun_out/miopen_db (about 4 minutes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
rm -rf gpurun_out/miopen_db && mkdir -p gpurun_out/miopen_db
python3 bench.py --no-cpu-baseline --conv-tuning search --conv-db gpurun_out/miopen_db > gpurun_out/tune_bench.json 2> gpurun_out/tune_bench.err
tail -3 gpurun_out/tune_bench.err; ls -la gpurun_out/miopen_db
