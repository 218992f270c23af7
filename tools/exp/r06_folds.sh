#!/bin/bash
# round 6: the small real files the way the reference runs them (cv -p on): one fold and five folds side by side
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_folds.jsonl; : > $O
for m in CAMF_C CAMF_CI BiasedMF CAMF_CU CAMF_CUCI; do
  for f in 1 5; do
    timeout 600 python bench.py --workload frappe --model $m --folds $f --steps 10 --warmup 2 2>/dev/null | tail -1 >> $O
  done
done
timeout 600 python bench.py --workload depaul --folds 5 --steps 10 --warmup 2 2>/dev/null | tail -1 >> $O
python - <<'PY'
import json
for l in open("gpurun_out/r06_folds.jsonl"):
    l=l.strip()
    if not l: continue
    d=json.loads(l); c=d["config"]
    print(c["workload"][:60], "| folds", c.get("concurrent_folds", c.get("folds")), "|", round(d["value"]/1e6,2), "M/s | cpu", d.get("cpu_baseline",{}).get("value") and round(d["cpu_baseline"]["value"]/1e6,2), d.get("cpu_baseline",{}).get("cores"))
PY
