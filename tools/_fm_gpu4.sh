python -m pytest tests/test_gpu_fm.py tests/test_reference_src_golden.py tests/test_gpu_dist_two_ranks.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_fm_tests2.log
cat gpurun_out/r4_fm_tests2.log
python bench.py --workload c4 --steps 5 --warmup 1 > gpurun_out/r4_c4_bench.json 2> gpurun_out/r4_c4_bench.err
cat gpurun_out/r4_c4_bench.json
tools/gpu_profile_fm.sh r04b > gpurun_out/r4_fm_prof2.log 2>&1
tail -25 gpurun_out/r4_fm_prof2.log
