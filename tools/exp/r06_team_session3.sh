#!/bin/bash
# round 6, GPU session 3: the soak test against a library WITHOUT the wait states (control), and what the wait states cost
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_team3
mkdir -p $O
export TMPDIR=/tmp
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_nofix.so timeout 900 python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | tail -40 ) > $O/soak_on_nofix_library.txt 2>&1
for lib in mi355x nofix; do
  for z in 1.1 0.8; do
    ( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_$lib.so timeout 900 python bench.py --workload small --item-zipf $z --steps 10 --warmup 2 --no-cpu-baseline --no-f64 --no-calibration 2>/dev/null | tail -1 ) > $O/bench_small_zipf${z}_$lib.json
    ( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_$lib.so timeout 900 python bench.py --workload small --item-zipf $z --steps 5 --warmup 2 --no-cpu-baseline --no-f64 --no-calibration --f64-primary 2>/dev/null | tail -1 ) > $O/bench_small_zipf${z}_f64_$lib.json
  done
done
grep -c . $O/*.json; tail -5 $O/soak_on_nofix_library.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_team3/bench_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], "%.1f M/s %.2f ms" % (d["value"]/1e6, d["ms_per_step"]), d["roofline"].get("schedule"), d["config"]["workload"][-40:])
    except Exception as e: print(f, "ERR", e)
PY
