cd /root/repo/scripts/micro
echo "== cold"; timeout 120 ./decode_chain 8 200 2>&1 | tail -6
echo "== DC_HOTW=1"; DC_HOTW=1 timeout 120 ./decode_chain 8 200 2>&1 | tail -6
echo "== DC_KSPLIT=2 cold"; DC_KSPLIT=2 timeout 120 ./decode_chain 8 200 2>&1 | tail -4
echo "== DC_KSPLIT=2 hot"; DC_KSPLIT=2 DC_HOTW=1 timeout 120 ./decode_chain 8 200 2>&1 | tail -4
