#!/bin/bash
# round 6: evalRankings with the selection confined to N compute units of every XCD and the contraction on the others (make EXP=1 library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_rank
mkdir -p $O
export TMPDIR=/tmp CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so
for n in 0 4 6 8 10 12 16; do
  ( CMI_RANK_SEL_CUS=$n timeout 600 python bench.py --workload rank --steps 8 --warmup 2 2>/dev/null | tail -1 ) > $O/rank_sel_cus_$n.json
done
python - <<'PY'
import json,glob,re
for f in sorted(glob.glob("gpurun_out/r06_rank/rank_sel_cus_*.json"), key=lambda x:int(re.findall(r"_(\d+)\.json",x)[0])):
    d=json.loads(open(f).read()); c=d["config"]
    print(f.split('/')[-1], "wall %.2f ms" % d["ms_per_step"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in c.items() if "ms" in k or "device" in k})
PY
