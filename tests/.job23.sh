set -e
mkdir -p gpurun_out/r03u
T="timeout -k 10 170 tests/fa_tune"
for v in 7 9 10; do $T 8 16 4096 128 1 --rounds 12 --only $v > gpurun_out/r03u/c_v$v.log 2>&1; done
for v in 3 5 6; do $T 8 16 4096 128 0 --rounds 12 --only $v > gpurun_out/r03u/nc_v$v.log 2>&1; done
$T 2 16 8192 128 1 --rounds 12 --only 7 > gpurun_out/r03u/c8192_v7.log 2>&1
$T 8 16 4096 128 1 --rounds 12 --only 7 --jpx 16 > gpurun_out/r03u/c_v7_half.log 2>&1
$T 8 16 4096 128 0 --rounds 12 --only 3 --jpx 16 > gpurun_out/r03u/nc_v3_half.log 2>&1
grep -h "core clock\|med " gpurun_out/r03u/*.log | cut -c1-200
