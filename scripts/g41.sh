#!/bin/bash
# A/B of the render kernels' load schedules (conv_lds, rconv_lds, gemm_tile: unconditional clamped loads, wave-parallel statistics merge,
# untracked L2 prefetch): libastts_base.so = the previous commit's build, libastts.so = this tree.  Parity tests first (new build).
cd "$GRAFT_REPO_ROOT"
L=autostyle-tts_amd/astts
timeout 1500 python -m pytest tests/test_synth_gpu.py tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -4
cp $L/libastts.so /tmp/new.so
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'seq', d['sequential_ms_per_step'], s)"; }
for i in 1 2 3; do
  cp $L/libastts_base.so $L/libastts.so; run base
  cp /tmp/new.so $L/libastts.so; run new
done
