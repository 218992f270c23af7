#!/bin/bash
# round 6, GPU session 2: the store-data hazard in isolation, and the owner epoch with / without the wait states
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_team2
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 tools/micro/bin/store_data_hazard 8 256 ) > $O/hazard_micro_8wps.jsonl 2>$O/hazard_micro_8wps.err
( timeout 300 tools/micro/bin/store_data_hazard 1 256 ) > $O/hazard_micro_1wps.jsonl 2>$O/hazard_micro_1wps.err
( TRACE_DIR=/tmp timeout 900 python tools/exp/owner_trace.py 64 default ) > $O/trace_k64_prefix.txt 2>&1
rm -f /tmp/trace_*.bin
export STATS_REPS=24
for lib in nofix exp b64; do
  for k in 64 128; do
    ( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_$lib.so timeout 900 python tools/exp/share_debug_stats.py $k ) > $O/stats_${lib}_k$k.txt 2>&1
  done
  ( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_$lib.so STATS_DTYPES=f64 timeout 900 python tools/exp/share_debug_stats.py 64 all ) > $O/stats_${lib}_k64_all.txt 2>&1
  ( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_$lib.so STATS_DTYPES=f64 timeout 900 python tools/exp/share_debug_stats.py 10 all ) > $O/stats_${lib}_k10_all.txt 2>&1
done
tail -n 12 $O/*.jsonl $O/stats*.txt; head -12 $O/trace_k64_prefix.txt; grep -A3 "self-inconsistent" $O/trace_k64_prefix.txt | head -30
