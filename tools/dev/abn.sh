#!/bin/bash
# bench.py on several builds of the library on ONE box, alternating, 3 rounds: tools/dev/abn.sh <lib> <lib> ... [-- bench.py arguments]
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2 3; do
  for L in "${LIBS[@]}"; do
    printf "%s " $(basename $L)
    MM_AMD_LIB=$L timeout 300 python3 bench.py --no-sharp "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['redo_utterances'])"
  done
done
