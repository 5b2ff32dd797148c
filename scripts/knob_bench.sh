#!/bin/bash
# bench.py (metric frame only, no cases, no CPU baseline) under a list of environment settings: "NAME=VALUE[,NAME=VALUE]" ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for setting in "$@"; do
  envs=$(echo "$setting" | tr ',' ' ')
  out=$(env $envs timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cases none --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['latency_ms_one_frame'], d['kernel_ms_per_frame_max_rank'])")
  echo "$setting: $out"
done
