#!/bin/bash
for l in 3 4 5 6; do python scripts/dev/conv_bench.py $l 2>/dev/null; done
