#!/bin/bash
# round 6, session 3: A/B of ranking-contraction variants built as one-off libraries under carskit_amd/lib/var/ (make OUT=... VARIANT=-D...):
# VARS="base st2 st3" tools/exp/r06_rank_variants.sh   (base = the shipped library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_rank_var; mkdir -p $O
for r in 1 2; do
  for v in ${VARS:-base}; do
    lib=carskit_amd/lib/var/lib$v.so; [ $v = base ] && lib=carskit_amd/lib/libcarskit_mi355x.so
    CMI_LIB_PATH=$PWD/$lib python bench.py --workload rank --steps 8 --warmup 2 > $O/${v}_$r.json 2> $O/${v}_$r.err
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_rank_var/*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']; c=d['config']
        print(f.split('/')[-1], 'wall %.3f dev %.3f gemm %.3f sel %.3f one-stream %.3f overl %s'%(d['ms_per_step'],c['device_ms_per_step'],r['kernel_ms'],r['selection']['kernel_ms'],r['device_ms_one_stream'],r['kernel_ms_while_overlapped']))
    except Exception as e: print(f, 'ERR', e)
P
