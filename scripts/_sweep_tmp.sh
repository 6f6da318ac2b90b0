mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_symmetry.py -q -x > gpurun_out/r05k/t2.log 2>&1; tail -3 gpurun_out/r05k/t2.log
python scripts/ab_time.py --libs ab/libogg_hip_r04.so ab/libogg_hip_sym1.so ocean_model_grid_generator_amd/csrc/libogg_hip.so --workloads r8 r16 r8_latdp r4_om4 --rounds 3 --extra="--launch pass" --json gpurun_out/r05k/ab3.json > gpurun_out/r05k/ab3.log 2>&1; grep -v "^{" gpurun_out/r05k/ab3.log | cut -c1-120 | tail -40
