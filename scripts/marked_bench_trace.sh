#!/bin/bash
# kernel trace of the loop bench (one stream, graph replays): per-kernel GPU time per step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-iou3d --steps 40 --warmup 5 "$@" 2>&1 | tail -1 | cut -c1-200
