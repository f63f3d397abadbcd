set -e
mkdir -p gpurun_out/r03B
T="timeout -k 10 250 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 21 --only 0,1,11,12 > gpurun_out/r03B/skipqk_ab.log 2>&1
grep -h "problem\|FAIL\| med \| ok" gpurun_out/r03B/*.log | cut -c1-220
