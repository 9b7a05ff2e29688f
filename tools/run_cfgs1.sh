# usage: bash tools/run_cfgs1.sh <lib> cfg...   (kernel time and integrator statistics of one library on several configurations)
A=$1; shift
for c in "$@"; do timeout -k 10 200 python tools/exp_time.py $c $A || exit 1; done
