#!/bin/bash
# Collect the judged profiles of bench.py for every single-GPU workload (run from the repo root through gpurun):
#   profiles/collect_all.sh [round tag, default r04]
# For each of cfg2 cfg2nc cfg1 cfg3: profiles/collect.sh <tag>_<workload> <workload>, then the three summaries are gathered
# under gpurun_out/profiles_<tag>/ with the names bench.py and DESIGN.md refer to:
#   <tag>_kernel_stats_<workload>.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --workload <workload> ...`
#   <tag>_hbm_traffic_<workload>.json   FETCH_SIZE / WRITE_SIZE (separate --pmc passes, FETCH doubled: gfx950) + csrc sha256
#   <tag>_sq_counters_<workload>.json   MFMA busy, clock, LDS bank conflicts
#   <tag>_kernel_timed_<workload>.json  the same trace, averaged over the timed region only (profiles/timed_region.py)
# Copy that directory's files into profiles/ and commit them.
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
for WL in cfg2 cfg2nc cfg1 cfg3; do
  bash $REPO/profiles/collect.sh ${TAG}_${WL} $WL || echo "collect failed for $WL"
  for KIND in kernel_stats.csv hbm_traffic.json sq_counters.json; do
    SRC=$REPO/gpurun_out/${TAG}_${WL}_${KIND}
    BASE=${KIND%.*}; EXT=${KIND##*.}
    [ -f $SRC ] && cp $SRC $OUT/${TAG}_${BASE}_${WL}.${EXT}
  done
  # the bench line printed under the --stats pass (its own HIP-event kernel time, to compare with the CSV's average)
  grep -h '^{' $REPO/gpurun_out/${TAG}_${WL}_stats.log | tail -1 > $OUT/${TAG}_bench_under_rocprof_stats_${WL}.json
  # average kernel duration over the bench's TIMED region (its last 20 dispatches) from the same trace
  (cd $REPO && python3 profiles/timed_region.py gpurun_out/${TAG}_${WL}_stats 20 > $OUT/${TAG}_kernel_timed_${WL}.json)
done
ls -la $OUT
