# usage: bash tools/exp/ab_flags_scan.sh REPS flag1 flag2 ... : default and every flag alternated REPS times, per-flag mean vs the mean of the defaults
export PYTHONPATH=$PWD
REPS=$1; shift
run() { python3 tools/exp/flag_time.py 30 $1 2>/dev/null | tail -1; }
for rep in $(seq 1 $REPS); do
  echo "default $(run "")"
  for f in "$@"; do echo "$f $(run "$f")"; done
done | tee /tmp/scan.txt
python3 - <<'PY'
import collections, statistics as st
d = collections.defaultdict(list)
for l in open("/tmp/scan.txt"):
    k, v = l.split()
    d[k].append(float(v))
base = st.mean(d["default"])
for k, v in d.items():
    print(f"{k:55s} {st.mean(v):.3f} ms  {100 * (st.mean(v) / base - 1):+.2f} %  (n={len(v)}, sd {st.pstdev(v):.3f})")
PY
