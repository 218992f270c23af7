#!/bin/bash
# usage (GPU box, via gpurun): tools/exp/fm_sq_profile.sh <tag> -- SQ / LDS counter passes of tools/bench_fm.py for the FM cell kernel (what do its waves wait for?)
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_${tag}_fmsq
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o pmc -- python tools/bench_fm.py 25000000 1 > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out="$out"
res={}
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
    for k,v in agg.items():
        for c,x in v.items(): res.setdefault(k,{})[c]=x/max(1,len(cnt[k]))
json.dump(res, open(out+"/sq.json","w"), indent=1)
for k,v in res.items():
    if "cell" in k or "reduce" in k: print(k, json.dumps(v))
PY
find $out -name "*kernel_trace.csv" -delete
