# Round profiles (run on the GPU box through gpurun): kernel stats of the benchmark command, PMC passes for HBM traffic and LDS conflicts.
# Usage: bash scripts/g_profiles.sh r06   (every profiler run is bounded by `timeout`: a hung timeout 600 rocprofv3 once cost a 20-minute call)
R=${1:-r06}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/p1 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch --no-side > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/${R}_bench_under_rocprof.err
python3 scripts/demangle_csv.py $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
# the decode step alone, one stream, nothing beside it: per-kernel durations free of the pipeline's contention (what `roofline.sequential` of the bench line reports)
rm -rf /tmp/p4 && PROBE_TS=250 PROBE_ITERS=2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o s --output-format csv -- python3 scripts/fullsize_probe.py > gpurun_out/${R}_sequential_probe.log 2>&1
python3 scripts/demangle_csv.py $(find /tmp/p4 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_sequential_kernel_stats.csv
rm -rf /tmp/p2 && PROBE_TS=12 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o p --output-format csv -- python3 scripts/fullsize_probe.py > /dev/null 2>&1
( echo "== timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 scripts/fullsize_probe.py (PROBE_TS=12 PROBE_ITERS=1); per kernel: (launches, mean FETCH_SIZE [KB] per launch); HBM bytes = value * 1024 * 2 on gfx950"; python scripts/pmc_summary.py /tmp/p2 gpurun_out/${R}_traffic_raw.json ) > gpurun_out/${R}_pmc_fetch_synth.txt
# the retrieval scan at the benchmark's shape (1000 x 6144 fp16 bank, Q = 8): its own FETCH_SIZE pass -> knn_roofline.traffic of the bench line
rm -rf /tmp/p5 && KNN_ITERS=50 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p5 -o k --output-format csv -- python3 scripts/knn_small.py > /dev/null 2>&1
( echo "== timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 scripts/knn_small.py (config-2 retrieval: N=1000, D=6144, Q=8, k=3)"; python scripts/pmc_summary.py /tmp/p5 gpurun_out/${R}_traffic_knn_raw.json ) > gpurun_out/${R}_pmc_fetch_knn.txt
# the 100k-bank retrieval stress of BASELINE configs[4] (bench.py knn_stress): Q = 8 (register-streaming scan), Q = 256 (the scan as one GEMM on the ring kernel), D = 768
: > gpurun_out/${R}_pmc_fetch_knn_stress.txt
for cfg in "100000 6144 8" "100000 6144 256" "100000 768 256"; do
  set -- $cfg
  rm -rf /tmp/p6 && KNN_N=$1 KNN_D=$2 KNN_Q=$3 KNN_ITERS=10 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p6 -o k --output-format csv -- python3 scripts/knn_small.py > /dev/null 2>&1
  ( echo "== timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 scripts/knn_small.py (N=$1, D=$2, Q=$3, k=3)"; python scripts/pmc_summary.py /tmp/p6 gpurun_out/${R}_traffic_knn_$1_$2_q$3.json ) >> gpurun_out/${R}_pmc_fetch_knn_stress.txt
done
python3 - <<PY
import json
t = json.load(open("gpurun_out/${R}_traffic_raw.json"))
k = json.load(open("gpurun_out/${R}_traffic_knn_raw.json"))
if "knn_scan" in k:
    t["knn_scan_1000_x_6144"] = k["knn_scan"]
for n, d, q in ((100000, 6144, 8), (100000, 6144, 256), (100000, 768, 256)):
    try:
        s = json.load(open(f"gpurun_out/${R}_traffic_knn_{n}_{d}_q{q}.json"))
    except OSError:
        continue
    fam = "knn_scan" if q < 64 else "gemm_ring"       # query groups of >= 64 take the scan as one GEMM on the ring kernel
    if fam in s:
        t[f"knn_scan_{n}_x_{d}_q{q}"] = s[fam]
json.dump(t, open("gpurun_out/${R}_traffic.json", "w"), indent=1)
print("traffic table:", {n: v["hbm_bytes_per_launch"] for n, v in t.items()})
PY
rm -rf /tmp/p3 && FLOW_N=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p3 -o p --output-format csv -- python3 scripts/flow_only.py > /dev/null 2>&1
( echo "== timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 scripts/flow_only.py (FLOW_N=1: warm-up + 1 solve); per kernel: (launches, mean counter value per launch)"; python scripts/pmc_summary.py /tmp/p3 ) > gpurun_out/${R}_pmc_flow_lds.txt
# the wide decode engine alone (128 rows x ~220 keys, 66 steps) and the flow solve at 64 sequences (32 utterances): what configs 3 / 4 / 5 run beside the headline's kernels
rm -rf /tmp/p7 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p7 -o w --output-format csv -- python3 scripts/wide_probe.py > gpurun_out/${R}_wide_probe.log 2>&1
python3 scripts/demangle_csv.py $(find /tmp/p7 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_wide_engine_kernel_stats.csv
rm -rf /tmp/p8 && FLOW_B=32 FLOW_N=2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o f --output-format csv -- python3 scripts/flow_only.py > gpurun_out/${R}_flow_b32.log 2>&1
python3 scripts/demangle_csv.py $(find /tmp/p8 -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_flow_b32_kernel_stats.csv
head -12 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-150; cat gpurun_out/${R}_pmc_fetch_synth.txt | head -12; cat gpurun_out/${R}_pmc_flow_lds.txt | head -8
