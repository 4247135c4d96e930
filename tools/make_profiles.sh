# Regenerate the committed profile artefacts of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/make_profiles.sh r04
# writes gpurun_out/prof_<round>_*; afterwards, in the build container:
#   python tools/profile_summary.py gpurun_out/prof_<round>_bf16x3 profiles/<round>_bench_1024_photo_bf16x3_kernel_stats.csv profiles/<round>_pmc_hbm_traffic.json
#   python tools/pmc_summary.py gpurun_out/pmc_fetch_<round> gpurun_out/pmc_write_<round> profiles/<round>_pmc_hbm_traffic.json "<cmd>" "<workload>" bf16x3 $(git rev-parse --short HEAD)
#   python tools/pmc_sq_summary.py gpurun_out/pmc_sq_<round> profiles/<round>_pmc_sq.json "<cmd>" "<workload>" bf16x3 $(git rev-parse --short HEAD)
# PMC counters are collected in their own passes, with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
R=${1:-r04}
B="python3 bench.py --no-cpu-baseline --no-extras --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_bf16x3 -- $B --steps 10 --warmup 2 > gpurun_out/prof_${R}_bf16x3.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_f16x2h -- $B --steps 10 --warmup 2 --precision f16x2h > gpurun_out/prof_${R}_f16x2h.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_${R} -- $B --steps 3 --warmup 1 > gpurun_out/pmc_fetch_${R}.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_${R} -- $B --steps 3 --warmup 1 > gpurun_out/pmc_write_${R}.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_${R} -- $B --steps 3 --warmup 1 > gpurun_out/pmc_sq_${R}.log 2>&1 &&
python3 bench.py > gpurun_out/bench_${R}.json 2> gpurun_out/bench_${R}.err &&
bash tools/cfg_bench.sh > gpurun_out/cfg_bench_${R}.jsonl 2>&1 &&
for A in "--frames 300" "--frames 300 --masked 0" "--frames 300 --png-level 1" "--frames 300 --streams 2" "--frames 300 --precision f16x2h"; do
  python3 tools/video_e2e.py $A 2> gpurun_out/video_e2e_${R}.err | grep "^{" >> gpurun_out/video_e2e_${R}.jsonl
done
