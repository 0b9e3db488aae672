cd $GRAFT_REPO_ROOT
ASTTS_BENCH_VERBOSE=1 timeout 900 python bench.py --no-24khz --no-cpu-baseline 2>&1 | grep -E "autotune|stream_pipe|Error|error|^\{" | cut -c1-300
