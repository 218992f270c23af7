#!/bin/bash
# round 6, GPU session 1 on the team-form deviation: all-owner trace + placement / coherence variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_team
O=gpurun_out/r06_team
export TMPDIR=/tmp
( TRACE_DIR=/tmp timeout 900 python tools/exp/owner_trace.py 64 default ) > $O/trace_k64.txt 2>&1
rm -f /tmp/trace_*.bin
export STATS_DTYPES=f64
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so timeout 600 python tools/exp/share_debug_stats.py 64 ) > $O/stats_base.txt 2>&1
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_hubcoh.so timeout 600 python tools/exp/share_debug_stats.py 64 ) > $O/stats_hubcoh.txt 2>&1
# CU-disjoint: the instance under test on the even compute units of every XCD pair..., neighbours on the others
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so STATS_CUS_TEST=16:0,1,2,3,4,5,6,7 STATS_CUS_NEIGH=16:8,9,10,11,12,13,14,15 STATS_NEIGH_WAVES=200 timeout 600 python tools/exp/share_debug_stats.py 64 ) > $O/stats_cu_disjoint.txt 2>&1
# XCD-disjoint (if mask bit i belongs to XCD i mod 8)
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so STATS_CUS_TEST=8:0,1,2,3 STATS_CUS_NEIGH=8:4,5,6,7 STATS_NEIGH_WAVES=200 timeout 600 python tools/exp/share_debug_stats.py 64 ) > $O/stats_xcd_disjoint.txt 2>&1
# same masks for everyone (control: masked streams, shared CUs)
( CMI_LIB_PATH=$PWD/carskit_amd/lib/libcarskit_exp.so STATS_CUS_TEST=8:0,1,2,3 STATS_CUS_NEIGH=8:0,1,2,3 STATS_NEIGH_WAVES=200 timeout 600 python tools/exp/share_debug_stats.py 64 ) > $O/stats_same_mask.txt 2>&1
tail -n 30 $O/*.txt
