#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_profile_fm.sh <tag> [ratings] -- kernel-trace stats + separate PMC passes of tools/bench_fm.py
tag=$1; n=${2:-25000000}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_${tag}_fm
mkdir -p $out
python tools/bench_fm.py $n 3 > $out/bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python tools/bench_fm.py $n 2 > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o pmc -- python tools/bench_fm.py $n 1 > $out/$c.log 2>&1
done
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $out/TCC -o pmc -- python tools/bench_fm.py $n 1 > $out/TCC.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/SQ -o pmc -- python tools/bench_fm.py $n 1 > $out/SQ.log 2>&1
python - <<PY
import csv, glob, collections, json
out="$out"
res={}
for name in ("FETCH_SIZE","WRITE_SIZE","TCC","SQ"):
    for f in glob.glob(out+"/"+name+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        seen=set()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
            key=(row["Dispatch_Id"],)
            if key not in seen:
                seen.add(key); cnt[k]+=1
        for k,v in agg.items():
            res.setdefault(k,{})["dispatches_"+name]=cnt[k]
            for c,x in v.items(): res[k][c+"_per_dispatch"]=x/max(1,cnt[k])
json.dump(res, open(out+"/pmc.json","w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
find $out -name "*kernel_trace.csv" -size +20M -delete
cat $out/bench.log | tail -2
cat $out/stats/*/*kernel_stats.csv 2>/dev/null | head -20 || find $out/stats -name "*stats*.csv" | head
