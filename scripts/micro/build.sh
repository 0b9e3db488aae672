#!/bin/sh
# Builds the stand-alone micro-benchmarks for gfx950 (run them on the GPU box: gpurun -- 'cd scripts/micro && ./ring_bench').
#   ring_bench          GEMM kernels on the flow decoder's projection shapes (ASTTS_GEMM_RING=0|1|2|3 selects the kernel / tile)
#   ring_bench_<parts>  the same with parts of gemm_ring compiled out, to see where a block's time goes
#   ring_bench_local    every block stores to the first rows: epilogue cost without HBM traffic
#   glds_check          semantics of global_load_lds_dwordx4 (lane-linear LDS destination)
#   coherence           reader-after-writer in one stream while a second stream keeps the GPU busy
set -e
cd "$(dirname "$0")"
F="-O3 --offload-arch=gfx950"
hipcc $F ring_bench.hip -o ring_bench
hipcc $F -DRING_SKIP_MFMA ring_bench.hip -o ring_bench_nomfma
hipcc $F -DRING_SKIP_EPI ring_bench.hip -o ring_bench_noepi
hipcc $F -DRING_SKIP_LOAD ring_bench.hip -o ring_bench_noload
hipcc $F -DRING_SKIP_MFMA -DRING_SKIP_EPI ring_bench.hip -o ring_bench_loadonly
hipcc $F -DEPI_DBG_LOCAL ring_bench.hip -o ring_bench_local
hipcc -O2 --offload-arch=gfx950 glds_check.hip -o glds_check
hipcc -O2 --offload-arch=gfx950 coherence.hip -o coherence
#   decode_chain        the LM decode step as a stand-alone launch chain (csrc/lm_step.hip): per-operator periods, eager vs
#                       hipGraph replay, one chain vs two concurrent ones; decode_chain_stamps: in-kernel s_memrealtime stamps
L="-mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16"      # as csrc/Makefile builds lm_step.hip
hipcc $F $L -std=c++17 decode_chain.hip -o decode_chain
hipcc $F $L -std=c++17 -DLM_STAMPS decode_chain.hip -o decode_chain_stamps
#   decode_chain_nt     weight lines loaded with the non-temporal policy (A/B of the guide's "nt-weights" row)
hipcc $F $L -std=c++17 -DLM_NT_WEIGHTS decode_chain.hip -o decode_chain_nt
#   cu_census           which CUs a hipExtStreamCreateWithCUMask stream really gets, per mask pattern
hipcc -O2 --offload-arch=gfx950 cu_census.hip -o cu_census
#   xlane_probe         csrc/xlane.h (DPP / v_permlane swaps) against __shfl_xor: bit equality and the latency of a 64-lane sum
hipcc $F -std=c++17 -Wno-unused-result xlane_probe.hip -o xlane_probe
