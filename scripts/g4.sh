cd $GRAFT_REPO_ROOT
X="DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 DC_BG=1"
for cfg in "127 96" "127 128" "127 160" "127 192" "100 160" "80 160" "80 128" "80 96"; do set -- $cfg; env $X DC_BG_LDS_KB=$1 DC_BG_NREG=$2 ./scripts/micro/decode_chain 8 100; done
