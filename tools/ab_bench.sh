# same-box A/B of two library builds:  tools/ab_bench.sh <lib A> <lib B> [pairs] [bench args...]
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq 1 $N); do
  for L in $A $B; do
    VSTNET_HIP_LIB=$L python bench.py --no-cpu-baseline --no-extras --steps 150 "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
    python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('$L', d['value'], d['ms_per_step'], flush=True)"
  done
done
