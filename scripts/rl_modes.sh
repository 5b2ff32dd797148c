#!/bin/bash
# traceRL on room.obj under each RT_RL_MODE bit set (diagnostics; see trace3_rl_kernel.hip)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for m in ${MODES:-0 1 2 3 4 7}; do
  RT_RL_MODE=$m timeout -k 10 120 python bench.py --rl --scene room --spp 4 --filter 0 --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode $m', d['ms_per_step'], 'ms', d['value'], 'Mrays/s', 'updates', d['q_updates_total'])" || exit 1
done
