#!/bin/bash
# GPU box: the rocprofv3 passes behind profiles/ (kernel trace of a bench run, then one counter pass per run,
# counters and traces never combined). Writes rocpd databases under gpurun_out/; summarise them here with
# tools/prof_summary.py and tools/pmc_summary.py.     bash tools/collect_profiles.sh TAG [bench.py arguments]
set -o pipefail
tag=${1:-x}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# the default run keeps its CPU baseline leg (the committed JSON line is the contract's); other workloads skip it: minutes
# of silent CPU work (the box ends a command that writes nothing for three minutes)
if [ $# -eq 0 ]; then cpu=""; else cpu="--cpu-spp 0"; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python3 bench.py "$@" $cpu > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}_bench.err || exit 1
echo "kernel trace done"
pass() {  # name, counters
  name=$1; counters=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc $counters -d gpurun_out/pmc_${tag}_$name -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-spp 0 "$@" > gpurun_out/pmc_${tag}_$name.json 2> gpurun_out/pmc_${tag}_$name.err || exit 1
  echo "pmc $name done"
}
pass issue "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "$@"
pass mix "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "$@"
pass fetch "FETCH_SIZE" "$@"
pass write "WRITE_SIZE" "$@"
# text summaries next to the databases; the databases themselves (tens of MB) are dropped so that gpurun_out/ stays small
python3 tools/prof_summary.py "$(ls gpurun_out/prof_$tag/*/*.db gpurun_out/prof_$tag/*.db 2>/dev/null | head -1)" gpurun_out/${tag}_kernel_stats.txt "kernel trace: python3 bench.py $*" > /dev/null || exit 1
for name in issue mix fetch write; do
  python3 tools/pmc_summary.py "$(ls gpurun_out/pmc_${tag}_$name/*/*.db gpurun_out/pmc_${tag}_$name/*.db 2>/dev/null | head -1)" gpurun_out/${tag}_pmc_$name.txt "counter pass $name: python3 bench.py --steps 1 --warmup 0 --cpu-spp 0 $*" > /dev/null || exit 1
done
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_${tag}_issue gpurun_out/pmc_${tag}_mix gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write
echo "summaries written"
