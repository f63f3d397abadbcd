#!/bin/bash
# Collect the judged profiles for one bench.py workload on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh <tag> [workload]          e.g.  profiles/collect.sh r01_v6 cfg2
# 1. rocprofv3 --kernel-trace --stats of the bench command      -> gpurun_out/<tag>_stats/
# 2. rocprofv3 --pmc FETCH_SIZE   (own pass, kernel trace only) -> gpurun_out/<tag>_fetch/
# 3. rocprofv3 --pmc WRITE_SIZE   (own pass)                    -> gpurun_out/<tag>_write/
# 4. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES (own pass)
# then profiles/summarize.py turns them into small files under gpurun_out/ to be copied into profiles/.
# The program itself follows `--` (no env / bash -c hop: the profiler initialises the GPU before the program starts).
set -e
mkdir -p ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out
TAG=${1:?tag}
WL=${2:-cfg2}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
# (--no-cfg4 --no-bf16-out: the trace then ends with the K timed steps of the workload itself: profiles/timed_region.py)
BENCH="python3 $REPO/bench.py --workload $WL --no-cpu-baseline --no-ceiling --no-cfg4 --no-bf16-out --steps 20 --warmup 5"
OUT=$REPO/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- $BENCH > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o run -- $BENCH > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -o run -- $BENCH > $OUT/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_sq -o run -- $BENCH > $OUT/${TAG}_sq.log 2>&1
cd $REPO && python3 profiles/summarize.py $TAG $WL
