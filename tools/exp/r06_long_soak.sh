#!/bin/bash
# round 6: a long soak of the shipped library (8x the test suite's repetitions) + 200 repetitions of the round-5 reproducer
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_long_soak; mkdir -p $O
( time CMI_SOAK_REPS=400 timeout 2400 python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | tail -4 ) > $O/soak_400_reps.txt 2>&1
( STATS_REPS=200 STATS_DTYPES=f64 timeout 1200 python tools/exp/share_debug_stats.py 64 all ) > $O/stats_k64_all_200.txt 2>&1
( STATS_REPS=200 STATS_DTYPES=f64 timeout 1200 python tools/exp/share_debug_stats.py 64 ) > $O/stats_k64_auto_200.txt 2>&1
( STATS_REPS=100 timeout 1200 python tools/exp/share_debug_stats.py 128 all ) > $O/stats_k128_all_100.txt 2>&1
( STATS_REPS=100 timeout 1200 python tools/exp/share_debug_stats.py 10 all ) > $O/stats_k10_all_100.txt 2>&1
( STATS_REPS=100 STATS_DTYPES=f32 timeout 1200 python tools/exp/share_debug_stats.py 256 all ) > $O/stats_k256_all_100.txt 2>&1
tail -n 4 $O/*.txt
