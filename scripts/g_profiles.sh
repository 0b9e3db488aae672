# Round profiles (run on the GPU box through gpurun): kernel stats of the benchmark command, PMC passes for HBM traffic and LDS conflicts.
# Usage: bash scripts/g_profiles.sh r03
R=${1:-r03}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/${R}_bench_under_rocprof.err
python3 scripts/demangle_csv.py $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
# the decode step alone, one stream, nothing beside it: per-kernel durations free of the pipeline's contention (what `roofline.sequential` of the bench line reports)
rm -rf /tmp/p4 && PROBE_TS=250 PROBE_ITERS=2 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o s --output-format csv -- python3 scripts/fullsize_probe.py > gpurun_out/${R}_sequential_probe.log 2>&1
python3 scripts/demangle_csv.py $(find /tmp/p4 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_sequential_kernel_stats.csv
rm -rf /tmp/p2 && PROBE_TS=12 PROBE_ITERS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p --output-format csv -- python3 scripts/fullsize_probe.py > /dev/null 2>&1
( echo "== rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 scripts/fullsize_probe.py (PROBE_TS=12 PROBE_ITERS=1); per kernel: (launches, mean FETCH_SIZE [KB] per launch); HBM bytes = value * 1024 * 2 on gfx950"; python scripts/pmc_summary.py /tmp/p2 gpurun_out/${R}_traffic_raw.json ) > gpurun_out/${R}_pmc_fetch_synth.txt
rm -rf /tmp/p3 && FLOW_N=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p3 -o p --output-format csv -- python3 scripts/flow_only.py > /dev/null 2>&1
( echo "== rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 scripts/flow_only.py (FLOW_N=1: warm-up + 1 solve); per kernel: (launches, mean counter value per launch)"; python scripts/pmc_summary.py /tmp/p3 ) > gpurun_out/${R}_pmc_flow_lds.txt
head -12 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-150; cat gpurun_out/${R}_pmc_fetch_synth.txt | head -12; cat gpurun_out/${R}_pmc_flow_lds.txt | head -8
