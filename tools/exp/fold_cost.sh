# What the split-K fold passes cost inside the replayed step (timing experiment, `make EXP=1` build under _exp_lib/): the step with
# every fold launch skipped (weight gradients garbage) against the full step, same box, alternating.
export MIRROR_HIP_LIB=$PWD/_exp_lib/libmirror_exp.so PYTHONPATH=$PWD
for i in 1 2 3; do
echo "full   $(python3 tools/exp/step_time.py 30 2>/dev/null | tail -1)"
echo "nofold $(MH_EXP_SKIP_FOLD=1 python3 tools/exp/step_time.py 30 2>/dev/null | tail -1)"
done
