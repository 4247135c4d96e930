for k in 20 200 1000 200; do
  python bench.py --no-cpu-baseline --no-extras --steps $k --warmup 5 --streams 3 > gpurun_out/st_$k.json 2>gpurun_out/st_$k.err || exit 1
  python -c "
import json
d=json.loads(open('gpurun_out/st_$k.json').read().strip().splitlines()[-1])
print('steps', $k, d['value'], d['ms_per_step'])"
done
