cd $GRAFT_REPO_ROOT
for i in 1 2; do
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain 8 200 2>&1 | head -1
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain_pl 8 200 2>&1 | head -1
done
DC_NOSTATE=1 DC_OPS=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain 8 100
DC_NOSTATE=1 DC_OPS=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain_pl 8 100
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain 16 100 | head -1
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain_pl 16 100 | head -1
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain 32 100 | head -1
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain_pl 32 100 | head -1
DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1 ./scripts/micro/decode_chain_pl 8 200 2>&1 | grep "TWO\|2 conc\|3 conc"
