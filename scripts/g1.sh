set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd scripts/micro && hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor ) > gpurun_out/launch_floor.log 2>&1
PROBE_ITERS=3 python scripts/fullsize_probe.py > gpurun_out/probe0.log 2>&1
rm -rf /tmp/tr && PROBE_ITERS=2 rocprofv3 --kernel-trace -d /tmp/tr -o tr --output-format csv -- python3 scripts/fullsize_probe.py > gpurun_out/probe_trace.log 2>&1
python scripts/step_timeline.py /tmp/tr > gpurun_out/step_timeline.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
cat gpurun_out/launch_floor.log gpurun_out/probe0.log gpurun_out/step_timeline.log
