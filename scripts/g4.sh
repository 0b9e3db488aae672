cd $GRAFT_REPO_ROOT
X="DC_NOSTATE=1 DC_KSPLIT=2 DC_LNPLAIN=1"
for b in 8 16 32; do for bin in decode_chain_f32 decode_chain; do env $X ./scripts/micro/$bin $b 150 | head -1; env $X DC_OPS=1 ./scripts/micro/$bin $b 100 | grep "wo "; done; done
