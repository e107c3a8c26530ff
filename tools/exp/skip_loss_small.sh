# in-step cost of ten of the loss section's tiny launches (their outputs feed only the heads' gradients)
run() { MH_EXP_SKIP=$1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-46s %8.3f ms' % ('$1'[:46], d['ms_per_step']))"; }
for i in 1 2 3; do run none; run mh_kl_fwd,mh_kl_bwd,mh_symkl_fwd,mh_symkl_bwd,mh_ce_rows_fwd,mh_ce_rows_bwd; done
