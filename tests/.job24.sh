set -e
mkdir -p gpurun_out/r03v
T="timeout -k 10 170 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 12 --only 7 > gpurun_out/r03v/c_v7.log 2>&1
$T 8 16 4096 128 0 --rounds 12 --only 3 > gpurun_out/r03v/nc_v3.log 2>&1
$T 4 8 2048 64 1 --rounds 12 --only 7 > gpurun_out/r03v/cfg1_v7.log 2>&1
$T 4 8 2048 64 1 --rounds 12 --only 1 > gpurun_out/r03v/cfg1_v1.log 2>&1
grep -h "core clock\|last launch\| med " gpurun_out/r03v/*.log | cut -c1-200
