#!/bin/bash
# round 6, session 3: A/B of chain-kernel build variants (one-off libraries under carskit_amd/lib/var/) on C3: VARS="base X" ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_c3_var; mkdir -p $O
for r in 1 2; do
  for v in ${VARS:-base}; do
    lib=carskit_amd/lib/var/lib$v.so; [ $v = base ] && lib=carskit_amd/lib/libcarskit_mi355x.so
    CMI_LIB_PATH=$PWD/$lib python bench.py --workload ${WL:-c3} --no-northstar --no-extras --no-f64 --no-cpu-baseline --steps 10 > $O/${v}_$r.json 2> $O/${v}_$r.err
    python - $O/${v}_$r.json $v <<'P'
import sys,json
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], '%.3f G/s  %.3f ms  frac %.4f  loss %.6f'%(d['value']/1e9,d['ms_per_step'],d['roofline']['frac'],d['config'].get('final_loss',0)))
except Exception as e: print(sys.argv[2],'ERR',e)
P
  done
done
