#!/bin/bash
for d in 0 1 2 4 6 8 14 15; do echo "== debug=$d"; GEECO_GEMM_DEBUG=$d python scripts/dev/conv_bench.py ${1:-3} 2>/dev/null | grep -E "fwd|dgrad"; done
