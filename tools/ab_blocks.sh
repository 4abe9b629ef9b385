#!/bin/bash
# A/B the chain variants at several workgroup sizes; one process per block size, variants interleaved inside
for B in ${BLOCKS:-1024 512 256}; do
  echo "== CVS_CHAIN_BLOCK=$B"
  CVS_CHAIN_BLOCK=$B timeout -k 10 300 python tools/ab_chain.py --variants "$1" --rounds 5 2>&1 | tail -n +1
done
