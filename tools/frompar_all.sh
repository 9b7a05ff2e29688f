#!/bin/bash
# The from-parameters leg of the bench line (parameters -> host tables -> handle -> cold step) for every configuration named.
#   gpurun --timeout 900 -- 'bash tools/frompar_all.sh r03l ncdm ncdm3 ...'  ->  gpurun_out/<tag>_frompar.txt
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
: > $ROOT/gpurun_out/${TAG}_frompar.txt
for c in "$@"; do
  timeout -k 10 200 python3 $ROOT/bench.py --config $c --no-cpu-baseline --no-secondary --steps 5 > $ROOT/gpurun_out/${TAG}_${c}_short.json 2> $ROOT/gpurun_out/${TAG}_${c}_short.err
  python3 - $c $ROOT/gpurun_out/${TAG}_${c}_short.json >> $ROOT/gpurun_out/${TAG}_frompar.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
s = d["stage_ms"]; f = s["from_parameters"]
print("%-16s step %.2f kernel %.2f cold %.2f | from parameters: host_tables %.2f create %.2f cold_step %.2f total %.2f parity %s" % (
    sys.argv[1], s["step_wall"], s["perturb_kernel"], s["cold_step_wall"], f.get("host_tables", -1), f.get("create_handle", -1), f.get("cold_step", -1),
    f.get("total", -1), f.get("parity", {}).get("ok") if "parity" in f else f))
PY
done
cat $ROOT/gpurun_out/${TAG}_frompar.txt
