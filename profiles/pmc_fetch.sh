#!/bin/bash
# usage: pmc_fetch.sh <tag> <bench args...>   -> gpurun_out/<tag>_fetch/
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/${TAG}_fetch -o run -- python3 $REPO/bench.py --no-cpu-baseline --no-ceiling --no-cfg4 --no-bf16-out --steps 10 --warmup 2 "$@" > $REPO/gpurun_out/${TAG}_fetch.log 2>&1
