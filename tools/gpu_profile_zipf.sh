#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_profile_zipf.sh <round tag> <item zipf exponent>
# The C3 shape with Zipf items (default schedule = the owner epoch): one bench line with the CPU baseline, the same without it under
# rocprofv3 --kernel-trace --stats, and the plain-level line (CMI_FLAG_NO_OWNER) for comparison.  Output: gpurun_out/prof_<tag>_zipf<z>/
tag=$1; z=$2
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_${tag}_zipf${z}
mkdir -p $out
CMI_OWNER_STATS=1 python bench.py --item-zipf $z --steps 3 --warmup 1 --no-f64 --no-calibration > $out/bench.json 2> $out/bench.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python bench.py --item-zipf $z --steps 3 --warmup 1 --no-cpu-baseline --no-f64 --no-calibration > $out/stats.log 2>&1
python bench.py --item-zipf $z --flags 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-f64 --no-calibration > $out/bench_levels.json 2> $out/bench_levels.err
find $out -name "*kernel_trace.csv" -size +20M -delete
grep "cmi\] owner" $out/bench.err | tail -1
cut -c1-300 $out/bench.json; cut -c1-200 $out/bench_levels.json
ls $out/stats
