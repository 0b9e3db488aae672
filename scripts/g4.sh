cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/trh && HIFT_N=2 rocprofv3 --kernel-trace -d /tmp/trh -o tr --output-format csv -- python3 scripts/hift_only.py > gpurun_out/hift_trace.log 2>&1
tail -1 gpurun_out/hift_trace.log
python scripts/trace_summary.py /tmp/trh 45 | grep -v "pack_weight\|pack_frag\|copyBuffer"
