#!/bin/bash
# Instruction mix of the perturbation kernel (rocprofv3 PMC passes, counters only + kernel trace):
#   gpurun -- 'bash tools/pmc_insts.sh <tag> <config>'  ->  gpurun_out/<tag>_inst_mix.txt
set -e -o pipefail
TAG=${1:-rXX}
CFG=${2:-lcdm}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/insts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --config $CFG > $OUT/$name.log 2>&1 || { echo "pass failed: $set"; tail -5 $OUT/$name.log; }
done
python3 - <<PY > $ROOT/gpurun_out/${TAG}_inst_mix.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_perturb" not in k and "k_los" not in k: continue
        acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s mean per launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $ROOT/gpurun_out/${TAG}_inst_mix.txt
