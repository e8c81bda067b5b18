#!/bin/bash
# tools/bench_big.py on several builds of the library on ONE box: tools/dev/abn_big.sh <lib> ... -- S P B N
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2; do
  for L in "${LIBS[@]}"; do
    printf "%s " $(basename $L)
    MM_AMD_LIB=$L timeout 300 python3 tools/bench_big.py "$@" 2>/dev/null | tail -1 | cut -c1-120
  done
done
