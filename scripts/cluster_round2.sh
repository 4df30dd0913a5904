#!/bin/bash
export TMPDIR=/tmp FHESTR_CLUSTER=1
O=gpurun_out
for v in ntx stag stagnt; do
  FHESTR_LIB=build/ab/libfhestr_$v.so timeout -k 10 300 python scripts/p44_bench.py --modes 1 8 32 256 > $O/c4_bench_$v.log 2>&1; echo $v; grep PBS $O/c4_bench_$v.log
done
export FHESTR_LIB=build/ab/libfhestr_stag.so
for B in 32; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
    tag=c4stag_b${B}_$(echo $C | cut -c1-11 | tr ' ' '_')
    timeout -k 10 200 bash scripts/pmc_cmd.sh $tag "$C" scripts/p44_prof.py $B > $O/pmc_$tag.txt 2>&1
    grep -A4 cluster $O/pmc_$tag.txt
  done
done
