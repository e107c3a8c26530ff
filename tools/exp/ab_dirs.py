#!/usr/bin/env python3
"""Interleaved A/B of two source trees (e.g. the working tree vs an exported base commit under _ab_base/) on the replayed c2 step:
ABBA order, N rounds, mean +- standard error of the paired difference (boxes drift by ~0.5 % within a minute).
usage: python3 tools/exp/ab_dirs.py [--rounds 4] [--steps 30] BASE_DIR VARIANT_DIR [bench args...]"""
import json, os, subprocess, sys, statistics
args = sys.argv[1:]
rounds, steps = 4, 30
while args and args[0] in ("--rounds", "--steps"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    else:
        steps = int(args[1])
    args = args[2:]
base, var, extra = os.path.abspath(args[0]), os.path.abspath(args[1]), args[2:]


def run(d):
    env = dict(os.environ, PYTHONPATH=d)
    out = subprocess.run([sys.executable, os.path.join(d, "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline"] + extra,
                         env=env, cwd=d, capture_output=True, text=True)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        sys.stderr.write(out.stderr[-2000:])
        raise


diffs, b_all, v_all = [], [], []
for r in range(rounds):
    order = (base, var, var, base) if r % 2 == 0 else (var, base, base, var)
    t = {base: [], var: []}
    for s in order:
        t[s].append(run(s))
    b, x = sum(t[base]) / 2, sum(t[var]) / 2
    diffs.append((x - b) / b * 100); b_all.append(b); v_all.append(x)
    print(f"round {r}: base {b:.3f} ms, variant {x:.3f} ms ({diffs[-1]:+.2f} %)", flush=True)
m = statistics.mean(diffs)
se = statistics.stdev(diffs) / len(diffs) ** 0.5 if len(diffs) > 1 else float("nan")
print(f"{var} vs {base}: step time {m:+.2f} % +- {se:.2f} (mean {statistics.mean(v_all):.3f} vs {statistics.mean(b_all):.3f} ms, {rounds} ABBA rounds)")
