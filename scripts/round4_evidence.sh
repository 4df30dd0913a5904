#!/bin/bash
# round-4 evidence in one GPU call: rocprofv3 stats + counters of the P22 step, of the P44 8-CU cluster kernel (256 LWEs) and
# of the whole-XCD kernel (16 LWEs: two clusters per XCD), the whole-XCD kernel's stamps, the parameter sweep
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 scripts/prof_round.py r04 > $O/prof_r04.log 2>&1; tail -5 $O/prof_r04.log
python3 scripts/prof_round.py r04_p44 --p44 > $O/prof_r04_p44.log 2>&1; tail -5 $O/prof_r04_p44.log
python3 scripts/prof_round.py r04_p44_xcd --batch=16 "--prog=python3 scripts/p44_prof.py 16" > $O/prof_r04_p44_xcd.log 2>&1; tail -5 $O/prof_r04_p44_xcd.log
for B in 8 16; do FHESTR_LIB=build/stamps/libfhestr_stamps.so timeout -k 10 200 python3 scripts/stamp_xcd.py $B; done > $O/r04_xcd_stamps.txt 2>&1; tail -3 $O/r04_xcd_stamps.txt
timeout -k 10 400 python3 scripts/p44_bench.py --modes 1,2 1 8 16 32 256 > $O/r04_p44_modes.txt 2>&1; tail -3 $O/r04_p44_modes.txt
timeout -k 10 400 python3 scripts/param_sweep.py 256 > $O/r04_param_sweep_b256.txt 2>&1; tail -3 $O/r04_param_sweep_b256.txt
