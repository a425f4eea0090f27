#!/bin/bash
# GPU box: FETCH_SIZE and WRITE_SIZE of tools/bin/fetch_calibrate (known byte counts), one counter per run.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 120 tools/bin/fetch_calibrate > gpurun_out/fetch_calibrate_plain.txt || exit 1
cat gpurun_out/fetch_calibrate_plain.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d gpurun_out/cal_$c -o pmc -- tools/bin/fetch_calibrate > gpurun_out/cal_$c.txt 2> gpurun_out/cal_$c.err || exit 1
  python3 tools/pmc_summary.py "$(ls gpurun_out/cal_$c/*/*.db gpurun_out/cal_$c/*.db 2>/dev/null | head -1)" gpurun_out/fetch_calibrate_$c.txt "tools/bin/fetch_calibrate, two dispatches of each kernel" > /dev/null || exit 1
  rm -rf gpurun_out/cal_$c
done
cat gpurun_out/fetch_calibrate_FETCH_SIZE.txt gpurun_out/fetch_calibrate_WRITE_SIZE.txt
