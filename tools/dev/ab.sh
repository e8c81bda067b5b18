#!/bin/bash
# A/B of two builds of the library on ONE box: tools/dev/ab.sh <lib A> <lib B> [bench.py arguments]   (alternating, 3 rounds)
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    printf "%s " $(basename $L)
    MM_AMD_LIB=$L timeout 300 python3 bench.py --no-sharp "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['redo_utterances'])"
  done
done
