#!/bin/bash
# LOCAL helper (build container): rebuild the HIP libraries if stale, stamp the commit, run one gpurun session.
#   tools/run_gpu.sh <timeout-seconds> '<command on the GPU box>'
cd "$(dirname "$0")/.."
python -m robo_amd.build > /tmp/robo_build.log 2>&1 || { tail -20 /tmp/robo_build.log; exit 1; }
git rev-parse --short HEAD > .git_head
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
