#!/bin/bash
# one GPU call: cluster-kernel tests, batch sweep of the default and nt-key builds, L2 / fabric counters at B = 8 and 32
export TMPDIR=/tmp FHESTR_CLUSTER=1
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_cluster.py -x -q > $O/c3_test.log 2>&1; tail -2 $O/c3_test.log
timeout -k 10 300 python scripts/p44_bench.py --modes 1 1 8 32 256 > $O/c3_bench.log 2>&1; grep PBS $O/c3_bench.log
FHESTR_LIB=build/ab/libfhestr_ntkey.so timeout -k 10 300 python scripts/p44_bench.py --modes 1 8 32 256 > $O/c3_bench_nt.log 2>&1; grep PBS $O/c3_bench_nt.log
for B in 8 32; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
    tag=c3_b${B}_$(echo $C | cut -c1-11 | tr ' ' '_')
    timeout -k 10 200 bash scripts/pmc_cmd.sh $tag "$C" scripts/p44_prof.py $B > $O/pmc_$tag.txt 2>&1
    grep -A4 cluster $O/pmc_$tag.txt
  done
done
