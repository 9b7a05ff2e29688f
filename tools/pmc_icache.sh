#!/bin/bash
# Instruction-cache behaviour of the perturbation kernel (rocprofv3 PMC passes, counters only + kernel trace):
#   gpurun -- 'bash tools/pmc_icache.sh <tag> <config> [lib]'  ->  gpurun_out/<tag>_icache.txt
set -e -o pipefail
TAG=${1:-rXX}
CFG=${2:-explanatory_mpk}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LIB=${3:-$ROOT/classpp_public_amd/csrc/libcpt.so}
OUT=$ROOT/gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/kernel_time.py $LIB $CFG > $OUT/$name.log 2>&1 || { echo "pass failed: $set"; tail -5 $OUT/$name.log; }
done
python3 - <<PY > $ROOT/gpurun_out/${TAG}_icache.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_perturb" not in k: continue
        acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean per launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $ROOT/gpurun_out/${TAG}_icache.txt
