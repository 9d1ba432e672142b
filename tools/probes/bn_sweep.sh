#!/bin/bash
# sweep of the grid caps of the BatchNorm passes (KP_BN_BLOCKS="fwd,reduce,apply") at model F's shapes
for caps in "0,1024,0" "4096,768,4096" "2048,512,2048" "1024,384,1024" "512,256,512"; do
  echo "== KP_BN_BLOCKS=$caps"; KP_BN_BLOCKS=$caps python tools/bench_bn.py 2>/dev/null | tail -10
done
