#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/collect_profiles.sh <tag> <workload>
# rocprofv3 kernel trace + stats of the default bench run, then FETCH_SIZE and WRITE_SIZE in separate counter passes;
# condensed into profiles/ by scripts/summarize_rocprof.py.
set -e
tag=$1; wl=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_${tag}_${wl}
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --workload $wl --steps 200 --warmup 3 --cpu-sample-div 0 --power-probe 0 --dp-arc-other 0 > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample-div 0 --power-probe 0 --dp-arc-other 0 --launch pass --tune-strips 0 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample-div 0 --power-probe 0 --dp-arc-other 0 --launch pass --tune-strips 0 > $out/write.log 2>&1
python3 scripts/summarize_rocprof.py $tag $wl $out/stats $out/fetch $out/write > $out/summary.txt
# the --stats table averages set-up, ramp and autotune launches with the timed ones: keep it as *_kernel_stats_all_*, and put the launches of
# the timed region (found by bench.py's clock stamps) into *_kernel_stats_*
mv profiles/${tag}_kernel_stats_${wl}.csv profiles/${tag}_kernel_stats_all_${wl}.csv
python3 scripts/trace_window.py $tag $wl $out/stats $out/stats.log >> $out/summary.txt
cp profiles/${tag}_* profiles/hbm_traffic.json $out/ 2>/dev/null || true
tail -20 $out/summary.txt
