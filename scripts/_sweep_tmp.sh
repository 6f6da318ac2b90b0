mkdir -p gpurun_out/r05c
export OGG_CAP_SYMMETRY=1
for w in 2 4 8; do
  for r in 0 $((w-1)); do
  echo "== r8 world $w rank $r"
  python scripts/env_sweep.py --workload r8 --as-world $w --as-rank $r --var OGG_PASS_LL_HELPERS --values 0 1 2 0 1 2 --steps 300
  v=OGG_PASS_LL_WG_SMALL; [ $w = 2 ] && v=OGG_PASS_LL_WG_MID
  python scripts/env_sweep.py --workload r8 --as-world $w --as-rank $r --var $v --values 60 90 120 180 240 60 90 120 180 240 --steps 300
  python scripts/env_sweep.py --workload r8 --as-world $w --as-rank $r --var OGG_PASS_ORDER --values 34201 01234 34201 01234 --steps 300
  python scripts/env_sweep.py --workload r8 --as-world $w --as-rank $r --var OGG_MESH_ROWS --values 2 4 8 2 4 8 --steps 300
  done
done > gpurun_out/r05c/sweep2.log 2>&1
echo "== r2 / r4_om4 helpers x wg"
for wl in r4_om4 r2; do
python scripts/env_sweep.py --workload $wl --var OGG_PASS_LL_WG_SMALL --set OGG_PASS_LL_HELPERS=2 --values 60 90 120 180 60 90 120 180 --steps 300
done >> gpurun_out/r05c/sweep2.log 2>&1
grep -v amdgpu.ids gpurun_out/r05c/sweep2.log | tail -200
