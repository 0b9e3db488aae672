#!/bin/bash
# round 4, session 2: driver tests, side workloads, default bench with the new cpu_baseline
mkdir -p gpurun_out
python -m pytest tests/test_configs_gpu.py::test_config3_longform_through_its_driver tests/test_cli_gpu.py tests/test_rccl_gpu.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -25
for w in config3 config5 config4; do
  echo "=== bench --workload $w"; timeout 900 python bench.py --workload $w --steps 2 --warmup 1 2> gpurun_out/r04_side_$w.err | tee gpurun_out/r04_side_$w.json | cut -c1-1500
  tail -3 gpurun_out/r04_side_$w.err
done
echo "=== default bench"; timeout 900 python bench.py 2> gpurun_out/r04_bench_a.err | tee gpurun_out/r04_bench_a.json | cut -c1-3000
tail -3 gpurun_out/r04_bench_a.err
