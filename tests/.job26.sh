set -e
mkdir -p gpurun_out/r03x
T="timeout -k 10 200 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 21 --only 1,3,4 > gpurun_out/r03x/queue_ab.log 2>&1
$T 2 16 8192 128 1 --rounds 11 --only 1,3,4 > gpurun_out/r03x/queue_ab_s8192.log 2>&1
$T 8 16 2048 128 1 --rounds 11 --only 1,3,4 > gpurun_out/r03x/queue_ab_s2048.log 2>&1
grep -h "problem\|FAIL\| med \| ok" gpurun_out/r03x/*.log | cut -c1-220
