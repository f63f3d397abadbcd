#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_tune -o run -- $REPO/tests/fa_tune 8 16 4096 128 0 --only 0,3,4 --rounds 3 > $REPO/gpurun_out/pmc_tune.log 2>&1
