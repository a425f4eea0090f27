#!/bin/bash
# GPU box: the rocprofv3 passes behind profiles/ (kernel trace of the default bench run, then one counter pass per
# run, counters and traces never combined). Writes rocpd databases under gpurun_out/; summarise them here with
# tools/prof_summary.py and tools/pmc_summary.py.     bash tools/collect_profiles.sh TAG
set -o pipefail
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python3 bench.py > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}_bench.err || exit 1
echo "kernel trace done"
pass() {  # name, counters
  timeout -s KILL 170 rocprofv3 --pmc $2 -d gpurun_out/pmc_${tag}_$1 -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-spp 0 > gpurun_out/pmc_${tag}_$1.json 2> gpurun_out/pmc_${tag}_$1.err || exit 1
  echo "pmc $1 done"
}
pass issue "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
pass mix "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
