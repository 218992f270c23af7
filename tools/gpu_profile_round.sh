#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_profile_round.sh <round tag, e.g. r02> <workload> [kernel substring]
# One bench line, a rocprofv3 kernel-trace summary and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the SAME bench.py
# command; results land in gpurun_out/prof_<tag>_<workload>/ and the summaries the judge reads are copied to profiles/ by hand.
# optional 4th argument: extra bench.py flags (e.g. --f64-primary); 5th: suffix of the output directory
tag=$1; wl=$2; kern=${3:-sgd_chain_level}; extra=${4:-}; sfx=${5:-}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_${tag}_${wl}${sfx}
mkdir -p $out
args=(--workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-f64 --no-calibration --no-northstar --no-extras $extra)
python bench.py "${args[@]}" > $out/bench.json 2> $out/bench.err
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python bench.py "${args[@]}" > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o pmc -- python bench.py "${args[@]}" > $out/$c.log 2>&1
done
timeout 1200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out/SQ -o pmc -- python bench.py "${args[@]}" > $out/SQ.log 2>&1
# translation / memory-side counters: one group per pass (skipped with PROFILE_QUICK=1)
if [ -z "$PROFILE_QUICK" ]; then
timeout 1200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --kernel-trace --output-format csv -d $out/UTCL -o pmc -- python bench.py "${args[@]}" > $out/UTCL.log 2>&1
timeout 1200 rocprofv3 --pmc TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum --kernel-trace --output-format csv -d $out/UTCL2 -o pmc -- python bench.py "${args[@]}" > $out/UTCL2.log 2>&1
timeout 1200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $out/TCC -o pmc -- python bench.py "${args[@]}" > $out/TCC.log 2>&1
timeout 1200 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum --kernel-trace --output-format csv -d $out/EA -o pmc -- python bench.py "${args[@]}" > $out/EA.log 2>&1
timeout 1200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace --output-format csv -d $out/TCP -o pmc -- python bench.py "${args[@]}" > $out/TCP.log 2>&1
fi
python tools/pmc_summary.py $out/FETCH_SIZE/pmc_counter_collection.csv $out/WRITE_SIZE/pmc_counter_collection.csv $kern $out/bench.json $out/pmc.json > $out/pmc_summary.log 2>&1
# keep only the per-kernel stats csv (the traces are large)
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out $out/stats | head -40
