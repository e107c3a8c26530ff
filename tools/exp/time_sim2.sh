#!/bin/bash
# where nys_sim2's ~60 us go: the file rebuilt with experiment hooks (results wrong), timed alone.  GPU box, repo root.
export PYTHONPATH=$PWD
for v in 0 1 2 3 4; do
  make -C mirror_amd/csrc -B build/nystrom_sim2.o FLAGS_nystrom_sim2="-DSIM2_EXP=$v" > /dev/null 2>&1 && make -C mirror_amd/csrc > /dev/null 2>&1
  echo -n "SIM2_EXP=$v: "; python3 tools/exp/time_sim2.py 2>/dev/null | tail -1
done
make -C mirror_amd/csrc -B build/nystrom_sim2.o > /dev/null 2>&1; make -C mirror_amd/csrc > /dev/null 2>&1
