#!/bin/bash
# bench.py ms_per_step of the metric frame under a list of environment settings (one per argument)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for setting in "$@"; do
  ( for kv in $setting; do export $kv; done
    timeout -k 10 200 python bench.py --cases none --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$setting]', d['ms_per_step'])" ) || exit 1
done
