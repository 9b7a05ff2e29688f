#!/bin/bash
# One tracked profile set: for every configuration named, the bench line (with the CPU baseline and the from-parameters / cold-step legs),
# the rocprofv3 kernel stats of the same workload and the FETCH_SIZE / WRITE_SIZE passes.
#   gpurun --timeout 1200 -- 'bash tools/profile_all.sh r03k explanatory_mpk ncdm ...'  ->  gpurun_out/<tag>_<config>_{bench.json,kernel_stats.csv,pmc_traffic.json}
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do
  echo "== $c" >&2
  python3 $ROOT/bench.py --config $c > $ROOT/gpurun_out/${TAG}_${c}_bench.json 2> $ROOT/gpurun_out/${TAG}_${c}_bench.err || { tail -5 $ROOT/gpurun_out/${TAG}_${c}_bench.err; exit 1; }
  bash $ROOT/tools/profile_bench.sh ${TAG}_$c $c
done
