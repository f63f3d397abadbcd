set -e
mkdir -p gpurun_out/r03A
T="timeout -k 10 250 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 25 --only 1,3,4,5,6 > gpurun_out/r03A/split_ab.log 2>&1
grep -h "problem\|FAIL\| med \| ok" gpurun_out/r03A/*.log | cut -c1-220
