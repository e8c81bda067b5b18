#!/bin/bash
# A/B of two environments on ONE box: tools/dev/ab_env.sh "<env A>" "<env B>" [bench.py arguments]   (alternating, 3 rounds)
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for E in "$A" "$B"; do
    printf "[%s] " "$E"
    env $E timeout 300 python3 bench.py --no-sharp "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['redo_utterances'])"
  done
done
