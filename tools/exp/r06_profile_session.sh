#!/bin/bash
# round 6: the profiles of the round -- default bench line; C3 kernel stats + PMC; C4 (FM, default = fixed-order sums) stats + PMC; rank
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PROFILE_QUICK=1
( timeout 1500 python bench.py > gpurun_out/r06_default_bench_line.json 2> gpurun_out/r06_default_bench_line.err )
bash tools/gpu_profile_round.sh r06 c3 sgd_chain_level > gpurun_out/r06_prof_c3.log 2>&1
bash tools/gpu_profile_fm.sh r06 > gpurun_out/r06_prof_fm.log 2>&1
bash tools/gpu_profile_aux.sh r06 > gpurun_out/r06_prof_aux.log 2>&1
ls gpurun_out/prof_r06_c3 gpurun_out/prof_r06_fm gpurun_out/prof_r06_c4 gpurun_out/prof_r06_rank 2>&1 | head -60
tail -c 1500 gpurun_out/r06_default_bench_line.json
