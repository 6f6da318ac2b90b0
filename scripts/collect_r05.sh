#!/bin/bash
# usage (GPU box, repo root): scripts/collect_r05.sh <part>     -- the round-5 evidence under profiles/r05_*, in parts of a few minutes each
#   a: r8 (kernel trace cut to the timed region, HBM traffic, VALU counters, bench line with the CPU baseline)
#   b: r8_latdp and r4_om4      c: r16
#   d: r2 bench line, the stencil pipeline (bench line, trace, VALU counters, cycle attribution), main() timing
#   e: rank sweeps (r8, r8_latdp, r16; 1 2 4 8 ranks)
#   f: the same workloads with every column evaluated (--cap-symmetry none): the bench lines beside the mirrored ones
part=$1
tag=r05
mkdir -p gpurun_out
case $part in
a) scripts/collect_round.sh $tag r8 ;;
b) scripts/collect_round.sh $tag r8_latdp r4_om4 ;;
c) scripts/collect_round.sh $tag r16 ;;
d)
  python3 bench.py --workload r2 --cpu-sample-div 0 > profiles/${tag}_bench_r2.json 2> gpurun_out/bench_${tag}_r2.err || exit 1
  python3 bench.py --workload r8 --latlon stencil --cpu-sample-div 0 --power-probe 0 > profiles/${tag}_bench_r8_stencil.json 2> gpurun_out/bench_${tag}_stencil.err || exit 1
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  out=gpurun_out/prof_${tag}_stencil; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --workload r8 --latlon stencil --steps 100 --warmup 3 --cpu-sample-div 0 --power-probe 0 > $out/stats.log 2>&1 || exit 1
  python3 scripts/trace_window.py $tag r8_stencil $out/stats $out/stats.log
  scripts/valu_counters.sh ${tag}_stencil r8 --latlon stencil > gpurun_out/valu_${tag}_stencil.log 2>&1 || exit 1
  scripts/stencil_counters.sh $tag > gpurun_out/stencil_counters_${tag}.log 2>&1
  python3 scripts/time_main.py profiles/${tag}_time_main.json > gpurun_out/time_main_${tag}.log 2>&1 || exit 1
  ;;
e)
  rm -f profiles/${tag}_rank_sweep.jsonl
  for wl in r8 r8_latdp r16; do for w in 1 2 4 8; do python3 scripts/rank_sweep.py --world $w --workload $wl --json profiles/${tag}_rank_sweep.jsonl 2>&1 | grep world; done; done
  ;;
f)
  for wl in r8 r8_latdp r4_om4 r16; do
    python3 bench.py --workload $wl --cap-symmetry none --cpu-sample-div 0 > profiles/${tag}_bench_${wl}_every_column.json 2> gpurun_out/bench_${tag}_${wl}_ec.err || exit 1
  done
  ;;
esac
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* profiles/hbm_traffic.json profiles/valu_counters.json gpurun_out/profiles_$tag/ 2>/dev/null
ls profiles/${tag}_* | wc -l
