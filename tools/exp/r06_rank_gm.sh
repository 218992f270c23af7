#!/bin/bash
# round 6, session 3: tile order of the ranking contraction -- super-rows of GM query bands (GM = 1: the old order, one band at a time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_rank_gm; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu ) > $O/tests.txt 2>&1; tail -n 1 $O/tests.txt
for r in 1 2; do
  for v in ${VARS:-gm1 gm4 gm8 gm16}; do
    lib=carskit_amd/lib/var/lib$v.so; [ $v = gm8 ] && lib=carskit_amd/lib/libcarskit_mi355x.so
    CMI_LIB_PATH=$PWD/$lib python bench.py --workload rank --steps 8 --warmup 2 > $O/${v}_$r.json 2> $O/${v}_$r.err
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_rank_gm/*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; c=d['config']
        print(f.split('/')[-1], 'wall %.3f dev %.3f gemm %.3f sel %.3f one-stream %.3f overl %s'%(d['ms_per_step'],c['device_ms_per_step'],r['kernel_ms'],r['selection']['kernel_ms'],r['device_ms_one_stream'],r['kernel_ms_while_overlapped']))
    except Exception as e: print(f, 'ERR', e)
P
