cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_rank2; export CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so
for p in none high low none high low; do
  if [ $p = none ]; then unset CMI_RANK_SEL_PRIO; else export CMI_RANK_SEL_PRIO=$p; fi
  (timeout 600 python bench.py --workload rank --steps 8 --warmup 2 2>/dev/null | tail -1) > gpurun_out/r06_rank2/prio_$p.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_rank2/prio_$p.json')); print('$p', round(d['config']['device_ms_per_step'],2), round(d['ms_per_step'],2))"
done
unset CMI_LIB_PATH CMI_RANK_SEL_PRIO
timeout 600 python -m pytest tests/test_gpu_owner.py -q -m gpu -k "lock_file or two_processes" 2>&1 | tail -3
