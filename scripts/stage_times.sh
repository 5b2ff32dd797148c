#!/bin/bash
# Development aid: per-stage kernel times of a frame under a list of RT_WF_* settings.
#   scripts/stage_times.sh scene "" "RT_WF_FILL=4" "RT_WF_REFILL=32 RT_WF_FILL=8"
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCENE=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  OUT=$REPO/gpurun_out/stage_${SCENE}_$i
  rm -rf "$OUT"; mkdir -p "$OUT"
  ( for kv in $setting; do export $kv; done
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $REPO/scripts/stage_frames.py $SCENE ${FRAMES:-8} > "$OUT/run.log" 2>&1 ) || { echo "run failed: $setting"; tail -5 "$OUT/run.log"; exit 1; }
  python3 $REPO/scripts/stage_times.py "$OUT" "$SCENE [$setting]"
done
