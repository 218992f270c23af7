#!/bin/bash
# round 6, session 3 (after the contraction's epilogue, the scalar selection cursor, the host tail): GPU suite, default bench line, ranking profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s3f; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
( time timeout 1500 python bench.py > $O/default_bench_line.json 2> $O/default_bench_line.err ) 2> $O/default_bench_time.txt
wl=rank; out=$PWD/gpurun_out/prof_r06c_rank; mkdir -p $out
args=(--workload $wl --steps 3 --warmup 1)
python bench.py "${args[@]}" > $out/bench.json 2> $out/bench.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python bench.py "${args[@]}" --no-cpu-baseline > $out/stats.log 2>&1
timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $out/MFMA -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/MFMA.log 2>&1
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/FETCH_SIZE -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/FETCH.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/WRITE_SIZE -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/WRITE.log 2>&1
find $out -name "*kernel_trace.csv" -size +20M -delete
head -4 $out/stats/stats_kernel_stats.csv | cut -c1-200
tail -c 600 $O/default_bench_line.json; cat $O/default_bench_time.txt
