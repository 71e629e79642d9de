#!/bin/bash
# Marginal cost of the stages with 20 batches in flight (results are meaningless in the ablated runs)
run() { v=$(env "$@" ZKE_BENCH_NOCHECK=1 python bench.py --steps 1500 --warmup 80 --no-cpu --no-saturated 2>/dev/null | grep -o '"value": [0-9.]*'); echo "$* $v"; }
run A=0
run ZKE_DEBUG_SKIP_RSA=1
run ZKE_DEBUG_PARSE_STOP=7
run ZKE_DEBUG_PARSE_STOP=7 ZKE_DEBUG_SKIP_RSA=1
run ZKE_DEBUG_PARSE_STOP=4
run ZKE_DEBUG_PARSE_STOP=1
run ZKE_NO_FUSE_CANON=1
