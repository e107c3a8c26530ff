for pr in 0 46 32 23 19 16; do
  if [ $pr = 0 ]; then echo "== rule"; python tools/bench_misc.py ppeg 2>/dev/null | grep "ppeg_fwd"; else echo "== MH_PPEG_PR=$pr"; MH_PPEG_PR=$pr python tools/bench_misc.py ppeg 2>/dev/null | grep "ppeg_fwd"; fi
done
