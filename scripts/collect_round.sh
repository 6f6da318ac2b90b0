#!/bin/bash
# usage (GPU box, repo root): scripts/collect_round.sh <tag> [workload ...]
# Everything the round's profiles/ needs for each workload: rocprofv3 --kernel-trace --stats of the default bench run, FETCH_SIZE /
# WRITE_SIZE passes (hbm_traffic), the VALU counter passes, and the bench line itself (CPU baseline included for r8 only).
tag=$1; shift
wls=${*:-r8}
for wl in $wls; do
  scripts/collect_profiles.sh $tag $wl > gpurun_out/collect_${tag}_${wl}.log 2>&1 || exit 1
  scripts/valu_counters.sh $tag $wl > gpurun_out/valu_${tag}_${wl}.log 2>&1 || exit 1
  div=0; [ "$wl" = "r8" ] && div=1
  python3 bench.py --workload $wl --cpu-sample-div $div > profiles/${tag}_bench_${wl}.json 2> gpurun_out/bench_${tag}_${wl}.err || exit 1
  echo "$wl done: $(python3 -c "import json;d=json.loads(open('profiles/${tag}_bench_${wl}.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'])")"
done
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* profiles/hbm_traffic.json profiles/valu_counters.json gpurun_out/profiles_$tag/
