set -e
mkdir -p gpurun_out/r03w
T="timeout -k 10 170 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 12 --only 7 > gpurun_out/r03w/c_v7.log 2>&1
$T 8 16 4096 128 0 --rounds 12 --only 3 > gpurun_out/r03w/nc_v3.log 2>&1
$T 8 16 4096 128 1 --rounds 12 --only 6 > gpurun_out/r03w/c_v6.log 2>&1
grep -h "per XCD (work\|core clock\|last launch\| med " gpurun_out/r03w/*.log | cut -c1-300
