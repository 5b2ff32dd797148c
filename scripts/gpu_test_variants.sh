#!/bin/bash
# The GPU parity suite under scheduling settings other than the defaults: the picture, the ids and the counters must not
# depend on them.  (1) what the launcher chooses for scenes beyond the L2s, forced onto the small test scenes;
# (2) early end of the walk phase for full waves, single-ray waves allowed, early refill.
set -eu
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
RT_WF_WALK_STOP=3 RT_WF_WALK_MIN=1 RT_WF_REFILL=32 RT_WF_COOP=1 RT_WF_PRIMARY_VIA_EXTEND=1 \
  timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread | tail -2
RT_WF_WALK_STOP=1 RT_WF_WALK_MIN=40 RT_WF_MIN_RAYS=1 RT_WF_REFILL=8 \
  timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread | tail -2
# (3) every extend stage in the row-mode kernel (one ray per 16 lanes; tests/test_gpu_timed_config.py compares the two
# kernels case by case, this runs the WHOLE parity suite on rt_wf_extend_rows_kernel)
RT_WF_ROWS=1 \
  timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread | tail -2
# (4) the -DRT_DEBUG_WALK build (make debugwalk): links and leaf ranges checked inside every kernel that walks the tree
RT_HIP_LIB=$PWD/raytracer_amd/lib/variants/librt_hip_debugwalk.so \
  timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread | tail -2
# (5) the packed-fp32 pair tests (opt-in) on every packed scene; (6) the contiguous block order of the filter passes everywhere
RT_WF_PAIRS=1 \
  timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread | tail -2
RT_FILTER_MAP=1 \
  timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread | tail -2
# (7) rt_filter never in its lane-private form (the next frame's last kernel then waits for every pass)
RT_FILTER_LANE=0 \
  timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread | tail -2
