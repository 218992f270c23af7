#!/bin/bash
# usage: tools/gpu_pmc_pass.sh <tag> <bench args...>   -- separate rocprofv3 PMC passes of one bench.py command (GPU box)
# writes gpurun_out/pmc_<tag>/<pass>/...csv ; one counter group per run (never combined with other trace domains)
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
run() { # name, counters...
  name=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o $name -- python bench.py "${BENCH_ARGS[@]}" > $out/$name.log 2>&1
  echo "$name exit $?" >> $out/status.txt
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
