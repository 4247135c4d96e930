for P in bf16x3 f16x2h; do
  for CFG in "--mode art --frames-per-gpu 4 --steps 20" "--mode art --steps 60" "--size 4096 --steps 6 --warmup 2" "--height 1080 --width 1920 --steps 60" "--height 1080 --width 1920 --masked 5 --steps 60" "--height 1080 --width 1920 --masked 5 --mask-kind noise --steps 60" "--size 256 --frames-per-gpu 16 --steps 40" "--host-pipeline 200 --steps 60" "--recompute-style --steps 60"; do
    python bench.py $CFG --precision $P --no-cpu-baseline --no-extras > gpurun_out/cfg_tmp.json 2> gpurun_out/cfg_tmp.err || { tail -3 gpurun_out/cfg_tmp.err; }
    python - "$P" "$CFG" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/cfg_tmp.json').read().strip().splitlines()[-1])
hp=d.get('host_pipeline',{}).get('value')
print(json.dumps({"precision":sys.argv[1],"cfg":sys.argv[2],"frames_per_s":d['value'],"ms_per_step":d['ms_per_step'],"host_pipeline":hp,"flags":d.get('fp16_range_flags',{}).get('value')}), flush=True)
PY
  done
done
