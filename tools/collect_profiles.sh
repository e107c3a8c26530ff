#!/bin/bash
# Evidence set for profiles/ (run on the GPU box from the repo root): bash tools/collect_profiles.sh <tag>
# bench line, rocprofv3 kernel stats of the same command, eager-mode per-stream / top-launch views, PMC HBM traffic,
# template config line + kernel stats, host-feed (PCIe-inclusive) lines, B = 32 line.
# Afterwards, HERE: cp gpurun_out/prof_<tag>/<tag>_* gpurun_out/prof_<tag>/pmc_*.json profiles/   (bench.py reads profiles/pmc_*.json and
# reports them as stale unless their csrc digest is the tree's: the two pmc files are part of the set)
set -u
TAG=${1:-r03_x}; PART=${2:-all}; R=$PWD; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT      # PART: core | rest | all (two calls fit gpurun's 20-minute limit)
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
if [ "$PART" != "rest" ]; then
python3 $R/bench.py > $OUT/${TAG}_c2_bench.json 2> $OUT/bench.err
echo "[collect] bench done"; 
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o r -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_c2_bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_c2_kernel_stats.csv
echo "[collect] kernel stats done"
rm -rf /tmp/p2; MIRROR_GRAPH=0 MIRROR_RNA_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
t=$(find /tmp/p2 -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then
  python3 $R/tools/prof_streams.py $t 14 > $OUT/${TAG}_c2_per_stream.txt 2>&1
  python3 $R/tools/prof_top.py $t 60 > $OUT/${TAG}_c2_top_launches.txt 2>&1
  python3 $R/tools/prof_timeline.py $t > $OUT/${TAG}_c2_timeline.txt 2>&1
fi
echo "[collect] eager trace done"
rm -rf /tmp/p3; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/p4; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ff=$(find /tmp/p3 -name "*counter_collection.csv" | head -1); fw=$(find /tmp/p4 -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 $R/tools/pmc_summary.py $ff $fw > $OUT/pmc_traffic.json
(cd $R && bash tools/pmc_mfma.sh > $OUT/pmc_mfma_busy.json 2>/dev/null)      # SQ_VALU_MFMA_BUSY_CYCLES per kernel, its own PMC pass, keyed by the csrc digest
echo "[collect] pmc done"
# one graph-replayed step, launch by launch (the fastest step of a traced bench run: the eager re-runs of the roofline leg are longer)
(cd $R && bash tools/trace_step_raw.sh ${TAG}_graph > /dev/null 2>&1 && python3 tools/prof_step_listing.py gpurun_out/${TAG}_graph_kernel_trace.csv fastest > $OUT/${TAG}_c2_step_listing_graph_replay_traced.txt && rm -f gpurun_out/${TAG}_graph_kernel_trace.csv)
echo "[collect] replayed-step listing done"
fi
if [ "$PART" = "core" ]; then ls -la $OUT; exit 0; fi
python3 $R/bench.py --config template --no-cpu-baseline > $OUT/${TAG}_template_bench.json 2>/dev/null
rm -rf /tmp/p5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o r -- python3 $R/bench.py --config template --steps 5 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/p5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_template_kernel_stats.csv
echo "[collect] template done"
python3 $R/bench.py --batch 32 --no-cpu-baseline > $OUT/${TAG}_c2_bench_B32.json 2>/dev/null
python3 $R/bench.py --feed host-bf16 --steps 20 --no-cpu-baseline > $OUT/${TAG}_c2_bench_feed_host_bf16.json 2>/dev/null
python3 $R/bench.py --feed host --steps 20 --no-cpu-baseline > $OUT/${TAG}_c2_bench_feed_host_f32.json 2>/dev/null
MIRROR_GRAPH=0 python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_c2_bench_eager_rna_graph.json 2>/dev/null
python3 $R/bench.py --config c4 --no-cpu-baseline > $OUT/${TAG}_c4_bench.json 2>/dev/null
python3 $R/bench.py --config c4 --batch 16 --no-cpu-baseline > $OUT/${TAG}_c4_bench_B16.json 2>/dev/null
(cd $R && bash tools/bench_world2_dryrun.sh) > $OUT/${TAG}_world2_gloo_dryrun.txt 2>&1
python3 $R/tools/bench_rna.py 2>/dev/null | tail -3 > $OUT/${TAG}_rna_branch_alone.txt
python3 $R/bench.py --precision fp8 --no-cpu-baseline > $OUT/${TAG}_c5_fp8_bench.json 2>/dev/null
python3 $R/tools/bench_chain.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_chain_and_attention_isolated.txt
python3 $R/tools/trace_gemms.py template 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_template_gemm_calls.txt
python3 $R/tools/trace_gemms.py c2 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_c2_gemm_calls.txt
ls -la $OUT
