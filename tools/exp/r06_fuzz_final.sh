#!/bin/bash
# round 6, session 3: the differential fuzzers on the round's FINAL library (after the ranking kernels' changes), fresh seeds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_fuzz_final; mkdir -p $O
( time timeout 1200 python tests/tools/fuzz_gpu.py 6000 707 ) > $O/fuzz_gpu.txt 2>&1 &
( time timeout 1200 python tests/tools/fuzz_rank_fm.py 1500 707 ) > $O/fuzz_rank_fm.txt 2>&1 &
( time timeout 1200 python tests/tools/fuzz_rank_split.py 1500 707 ) > $O/fuzz_rank_split.txt 2>&1 &
( time timeout 1200 python tests/tools/fuzz_svdpp.py 200 707 ) > $O/fuzz_svdpp.txt 2>&1 &
wait
tail -n 5 $O/*.txt
