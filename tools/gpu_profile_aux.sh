#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_profile_aux.sh <round tag>   -- FM (C4 share) and evalRankings: one bench line each + the
# rocprofv3 kernel-trace summary of the same command (+ MFMA / LDS counters for the ranking contraction); results in
# gpurun_out/prof_<tag>_{c4,rank}/, summaries copied to profiles/ by hand.
tag=$1
export TMPDIR=/tmp
for wl in c4 rank; do
  out=$PWD/gpurun_out/prof_${tag}_${wl}
  mkdir -p $out
  args=(--workload $wl --steps 3 --warmup 1)
  python bench.py "${args[@]}" > $out/bench.json 2> $out/bench.err
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python bench.py "${args[@]}" --no-cpu-baseline > $out/stats.log 2>&1
  if [ $wl = rank ]; then
    timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $out/MFMA -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/MFMA.log 2>&1
    timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/FETCH_SIZE -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/FETCH.log 2>&1
    timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/WRITE_SIZE -o pmc -- python bench.py "${args[@]}" --no-cpu-baseline > $out/WRITE.log 2>&1
  fi
  find $out -name "*kernel_trace.csv" -size +20M -delete
done
ls -la gpurun_out/prof_${tag}_c4 gpurun_out/prof_${tag}_rank
