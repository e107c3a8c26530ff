# usage: bash tools/exp/ab_flags_n.sh N "flag" : default and the flag setting alternated N times (ABBA order), mean and standard error
export PYTHONPATH=$PWD
N=$1; F=$2
run() { python3 tools/exp/flag_time.py 30 $1 2>/dev/null | tail -1; }
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then a=$(run ""); b=$(run "$F"); else b=$(run "$F"); a=$(run ""); fi
  echo "round $i: default $a ms, $F $b ms"
done | tee /tmp/abn.txt
python3 - <<'PY'
import re, statistics as st
d = [(float(m.group(1)), float(m.group(2))) for m in (re.search(r"default ([\d.]+) ms, \S+ ([\d.]+) ms", l) for l in open("/tmp/abn.txt")) if m]
r = [100 * (b / a - 1) for a, b in d]
print(f"flag vs default: {st.mean(r):+.2f} % +- {st.stdev(r) / len(r) ** 0.5:.2f} ({len(r)} rounds; default mean {st.mean(a for a, _ in d):.3f} ms)")
PY
