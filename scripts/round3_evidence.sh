#!/bin/bash
# round-3 evidence in one GPU call: rocprofv3 stats + counters of the P22 step and of the P44 cluster kernel, the
# parameter sweep, the noise measurement of every kernel family
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 scripts/prof_round.py r03 > $O/prof_r03.log 2>&1; tail -5 $O/prof_r03.log
python3 scripts/prof_round.py r03_p44 --p44 > $O/prof_r03_p44.log 2>&1; tail -5 $O/prof_r03_p44.log
timeout -k 10 400 python3 scripts/param_sweep.py 256 > $O/r03_param_sweep_b256.txt 2>&1; tail -3 $O/r03_param_sweep_b256.txt
timeout -k 10 400 python3 scripts/noise_budget.py all 4096 > $O/r03_noise_all.log 2>&1; tail -1 $O/r03_noise_all.log | cut -c1-200
