#!/bin/bash
# bench.py with cases (default: room) under a list of environment settings, alternating, ROUNDS times:
#   scripts/knob_cases.sh "NAME=VALUE[,NAME=VALUE]" ...
# prints: metric frame ms (in flight), its latency alone, then per case ms in flight / latency alone
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for i in $(seq ${ROUNDS:-3}); do
  for setting in "$@"; do
    envs=$(echo "$setting" | tr ',' ' ')
    out=$(env $envs timeout -k 10 200 python bench.py --cases ${CASES:-room} --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('cases') or {}
print(d['ms_per_step'], d['latency_ms_one_frame'], *[f\"{k} {v['ms_per_frame']} {v['latency_ms_one_frame']}\" for k,v in c.items()])") || exit 1
    echo "$i $setting: $out"
  done
done
