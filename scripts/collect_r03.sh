#!/bin/bash
# usage (GPU box, repo root): scripts/collect_r03.sh <part>     -- the round-3 evidence under profiles/r03_*, in parts of a few minutes each
#   a: r8 (kernel stats, HBM traffic, VALU counters, bench line with the CPU baseline)
#   b: r8_latdp and r4_om4 (the same; literal arc form, the default)
#   c: r16 and r2 bench lines; the stencil pipeline (bench line, kernel stats, counters); main() timing
#   d: rank sweeps (r8, r8_latdp literal, r16), microbenchmarks, the literal quadrature's cycle profile
part=$1
tag=r03
mkdir -p gpurun_out
case $part in
a) scripts/collect_round.sh $tag r8 ;;
b) scripts/collect_round.sh $tag r8_latdp r4_om4 ;;
c)
  for wl in r16 r2; do python3 bench.py --workload $wl --cpu-sample-div 0 > profiles/${tag}_bench_${wl}.json 2> gpurun_out/bench_${tag}_${wl}.err || exit 1; done
  python3 bench.py --workload r8 --latlon stencil --cpu-sample-div 0 --power-probe 0 > profiles/${tag}_bench_r8_stencil.json 2> gpurun_out/bench_${tag}_stencil.err || exit 1
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  out=gpurun_out/prof_${tag}_stencil; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --workload r8 --latlon stencil --steps 100 --warmup 3 --cpu-sample-div 0 --power-probe 0 > $out/stats.log 2>&1 || exit 1
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats_r8_stencil.csv
  scripts/valu_counters.sh ${tag}_stencil r8 --latlon stencil > gpurun_out/valu_${tag}_stencil.log 2>&1 || exit 1
  python3 scripts/time_main.py profiles/${tag}_time_main.json > gpurun_out/time_main_${tag}.log 2>&1 || exit 1
  ;;
d)
  rm -f profiles/${tag}_rank_sweep.jsonl
  for wl in r8 r8_latdp r16; do for w in 1 2 4 8; do python3 scripts/rank_sweep.py --world $w --workload $wl --json profiles/${tag}_rank_sweep.jsonl 2>&1 | grep world; done; done
  { echo "== scripts/microbench/launch_overhead"; ./scripts/microbench/launch_overhead; echo "== scripts/microbench/tables_probe"; ./scripts/microbench/tables_probe; echo "== scripts/microbench/horner_issue"; ./scripts/microbench/horner_issue; } > profiles/${tag}_microbench.txt 2>&1
  OGG_LIB_PATH=$PWD/ab/libogg_hip_prof.so python3 scripts/dq_profile_run.py r8_latdp r4_om4 2>&1 | grep -A1 -E "^==|dq profile" | grep -v "^--" > profiles/${tag}_dq_profile.txt
  ;;
esac
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* profiles/hbm_traffic.json profiles/valu_counters.json gpurun_out/profiles_$tag/ 2>/dev/null
ls profiles/${tag}_* | wc -l
