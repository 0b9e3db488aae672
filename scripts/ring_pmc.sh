cd /root/repo
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pr$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pr$i -o p --output-format csv -- python3 scripts/ring_pmc.py > /dev/null 2>&1
  echo "== $grp"; python scripts/pmc_summary.py /tmp/pr$i | grep -i "gemm_ring"
done
rm -rf /tmp/prs; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prs -o p --output-format csv -- python3 scripts/ring_pmc.py > /dev/null 2>&1
head -5 /tmp/prs/*/*kernel_stats.csv 2>/dev/null || find /tmp/prs -name "*stats*" | head
