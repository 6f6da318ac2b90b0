#!/bin/bash
# usage (GPU box, repo root): scripts/collect_r04.sh <part>     -- the round-4 evidence under profiles/r04_*, in parts of a few minutes each
#   a: r16 (kernel stats, HBM traffic, VALU counters, bench line) -- BASELINE config 5, which round 3 left without counters
#   b: r8 (the same + the CPU baseline)
#   c: r8_latdp and r4_om4 in the default (chord) arc form, then their literal-form bench lines
#   d: the stencil pipeline (bench line, kernel stats, counters), r2 bench line, main() timing
#   c2: VALU counters of the literal displaced-pole form
#   e: rank sweeps (r8, r8_latdp, r16)
part=$1
tag=r04
mkdir -p gpurun_out
case $part in
a) scripts/collect_round.sh $tag r16 ;;
b) scripts/collect_round.sh $tag r8 ;;
c)
  scripts/collect_round.sh $tag r8_latdp r4_om4 || exit 1
  for wl in r8_latdp r4_om4; do python3 bench.py --workload $wl --dp-arc literal --cpu-sample-div 0 > profiles/${tag}_bench_${wl}_literal.json 2> gpurun_out/bench_${tag}_${wl}_literal.err || exit 1; done
  ;;
c2)  # counters of the literal arc form (opt-in since round 4), then its bench line again so that it quotes them
  scripts/valu_counters.sh ${tag}_literal r8_latdp --dp-arc literal > gpurun_out/valu_${tag}_literal.log 2>&1 || exit 1
  python3 bench.py --workload r8_latdp --dp-arc literal --cpu-sample-div 0 > profiles/${tag}_bench_r8_latdp_literal.json 2> gpurun_out/bench_${tag}_r8_latdp_literal.err || exit 1
  ;;
d)
  python3 bench.py --workload r2 --cpu-sample-div 0 > profiles/${tag}_bench_r2.json 2> gpurun_out/bench_${tag}_r2.err || exit 1
  python3 bench.py --workload r8 --latlon stencil --cpu-sample-div 0 --power-probe 0 > profiles/${tag}_bench_r8_stencil.json 2> gpurun_out/bench_${tag}_stencil.err || exit 1
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  out=gpurun_out/prof_${tag}_stencil; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --workload r8 --latlon stencil --steps 100 --warmup 3 --cpu-sample-div 0 --power-probe 0 > $out/stats.log 2>&1 || exit 1
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_kernel_stats_r8_stencil.csv
  scripts/valu_counters.sh ${tag}_stencil r8 --latlon stencil > gpurun_out/valu_${tag}_stencil.log 2>&1 || exit 1
  python3 scripts/time_main.py profiles/${tag}_time_main.json > gpurun_out/time_main_${tag}.log 2>&1 || exit 1
  ;;
e)
  rm -f profiles/${tag}_rank_sweep.jsonl
  for wl in r8 r8_latdp r16; do for w in 1 2 4 8; do python3 scripts/rank_sweep.py --world $w --workload $wl --json profiles/${tag}_rank_sweep.jsonl 2>&1 | grep world; done; done
  ;;
esac
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* profiles/hbm_traffic.json profiles/valu_counters.json gpurun_out/profiles_$tag/ 2>/dev/null
ls profiles/${tag}_* | wc -l
