#!/bin/bash
# usage (GPU box, via gpurun): tools/exp/rank_select_profile.sh  -- kernel-trace summary and SQ counters of the ranking bench's kernels
# for the tile form and the slab form of the selection; results under gpurun_out/rank_sel/
export TMPDIR=/tmp
out=$PWD/gpurun_out/rank_sel
mkdir -p $out
args=(--workload rank --steps 3 --warmup 1 --no-cpu-baseline)
for form in tile slab; do
  [ $form = slab ] && export CMI_RANK_NO_TILE=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${form}_stats -o s -- python bench.py "${args[@]}" > $out/${form}_stats.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/${form}_pmc -o p -- python bench.py "${args[@]}" > $out/${form}_pmc.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $out/${form}_pmc2 -o p -- python bench.py "${args[@]}" > $out/${form}_pmc2.log 2>&1
done
find $out -name "*kernel_trace.csv" -size +20M -delete
for form in tile slab; do
  echo "== $form"; f=$(find $out/${form}_stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
done
python - "$out" <<'PY'
import sys, glob, pandas as pd
out = sys.argv[1]
for form in ("tile", "slab"):
    for d in ("pmc", "pmc2"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, form, d), recursive=True):
            df = pd.read_csv(f)
            df = df[df.Kernel_Name.str.contains("rank_topn")]
            g = df.groupby(["Kernel_Name", "Counter_Name"]).Counter_Value.mean().unstack()
            g.index = [k[:40] for k in g.index]
            print("==", form, d); print(g.T.to_string())
PY
