set -e
mkdir -p gpurun_out/r03t
T="timeout -k 10 170 tests/fa_tune"
$T 8 16 4096 128 1 --rounds 15 --only 0,1,6,7 > gpurun_out/r03t/early_ab.log 2>&1
$T 8 16 4096 128 1 --rounds 5 --only 8 > gpurun_out/r03t/early_stamp.log 2>&1
$T 8 16 4096 128 1 --rounds 5 --only 10 > gpurun_out/r03t/prod_stamp_c.log 2>&1
$T 8 16 4096 128 0 --rounds 5 --only 3 > gpurun_out/r03t/prod_stamp_nc.log 2>&1
$T 32 16 2048 128 0 --rounds 5 --only 3 > gpurun_out/r03t/stamp_nc_s2048.log 2>&1
$T 2 16 8192 128 1 --rounds 5 --only 10 > gpurun_out/r03t/stamp_c_s8192.log 2>&1
$T 32 16 2048 128 1 --rounds 5 --only 10 > gpurun_out/r03t/stamp_c_s2048.log 2>&1
grep -h "med \|FAIL" gpurun_out/r03t/early_ab.log
