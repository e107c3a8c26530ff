#!/usr/bin/env python3
"""Interleaved A/B of environment switches on the replayed c2 step with statistics: ABBA order, N rounds, mean +- standard error of the
paired difference (boxes drift by ~0.5 % within a minute: single runs cannot resolve +-0.5 % effects).
usage: python3 tools/exp/ab_stat.py [--rounds 8] BASE_ENV VARIANT_ENV [VARIANT_ENV ...]      (each "K=V" or "K=V+K2=V2"; "-" = no change)"""
import json, os, subprocess, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
rounds = 8
if args and args[0] == "--rounds":
    rounds = int(args[1]); args = args[2:]
def run(spec):
    env = dict(os.environ, PYTHONPATH=R)
    if spec != "-":
        for kv in spec.split("+"):
            k, v = kv.split("="); env[k] = v
    out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], env=env, cwd="/tmp",
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)["ms_per_step"]
base, variants = args[0], args[1:]
for v in variants:
    diffs, b_all, v_all = [], [], []
    for r in range(rounds):
        order = (base, v, v, base) if r % 2 == 0 else (v, base, base, v)
        t = {base: [], v: []}
        for s in order:
            t[s].append(run(s))
        b, x = sum(t[base]) / 2, sum(t[v]) / 2
        diffs.append((x - b) / b * 100); b_all.append(b); v_all.append(x)
    m = statistics.mean(diffs); se = statistics.stdev(diffs) / len(diffs) ** 0.5
    print(f"{v:40s} vs {base}: step time {m:+.2f} % +- {se:.2f} (mean {statistics.mean(v_all):.3f} vs {statistics.mean(b_all):.3f} ms, {rounds} ABBA rounds)", flush=True)
