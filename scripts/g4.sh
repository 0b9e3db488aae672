cd /root/repo/scripts/micro
for pf in 0 1 0 1; do echo "== DC_PF=$pf"; DC_PF=$pf DC_KSPLIT=2 GPU_MAX_HW_QUEUES=8 timeout 200 ./decode_chain 8 200 2>&1 | grep -E "eager:|eager, TWO|eager, 2 conc" | head -3; done
