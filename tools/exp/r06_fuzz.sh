#!/bin/bash
# round 6: the differential fuzzers on the round's final library (seeds not used in earlier rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_fuzz; mkdir -p $O
( time timeout 1500 python tests/tools/fuzz_gpu.py 12000 606 ) > $O/fuzz_gpu.txt 2>&1 &
( time timeout 1500 python tests/tools/fuzz_rank_fm.py 1500 606 ) > $O/fuzz_rank_fm.txt 2>&1 &
( time timeout 1500 python tests/tools/fuzz_rank_split.py 600 606 ) > $O/fuzz_rank_split.txt 2>&1 &
( time timeout 1500 python tests/tools/fuzz_svdpp.py 400 606 ) > $O/fuzz_svdpp.txt 2>&1 &
wait
tail -n 6 $O/*.txt
