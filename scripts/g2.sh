cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/trf && FLOW_N=2 rocprofv3 --kernel-trace -d /tmp/trf -o tr --output-format csv -- python3 scripts/flow_only.py > gpurun_out/flow_trace.log 2>&1
python scripts/trace_summary.py /tmp/trf > gpurun_out/flow_summary.log 2>&1
tail -3 gpurun_out/flow_trace.log; cat gpurun_out/flow_summary.log
