#!/bin/bash
# bench.py under several environments on ONE box, alternating, 3 rounds: tools/dev/ab_envn.sh "<env A>" "<env B>" ... [-- bench.py arguments]
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2 3; do
  for E in "${ENVS[@]}"; do
    printf "[%s] " "$E"
    env $E timeout 300 python3 bench.py --no-sharp "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['redo_utterances'])"
  done
done
