python -m pytest tests/test_gpu_model_io.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r4_modelio.log
cat gpurun_out/r4_modelio.log
python -m pytest tests/test_gpu_ranking.py tests/test_reference_rank_golden.py tests/test_gpu_realdata.py -x -q -m gpu 2>&1 | tail -8
python bench.py --workload rank --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4_rank_bench.json 2> gpurun_out/r4_rank_bench.err
python -c "import json; d=json.load(open('gpurun_out/r4_rank_bench.json')); print(d['ms_per_step'], d['config']['device_ms_per_step'], d['config']['host_ms_breakdown_last_step'])"
