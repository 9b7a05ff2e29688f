#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs
# (MI355X_MICROARCH.md: counters in separate --pmc passes, never combined with the hip/hsa trace domains).
#   gpurun -- 'bash tools/profile_bench.sh r01c'   ->  gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.json
set -e -o pipefail
TAG=${1:-rXX}
CFG=${2:-lcdm}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-from-parameters --config $CFG > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-from-parameters --config $CFG > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-from-parameters --config $CFG > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_summarise.py $OUT $ROOT/gpurun_out/$TAG
