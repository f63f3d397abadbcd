set -e
mkdir -p gpurun_out/r03z
T="timeout -k 10 170 tests/fa_tune"
tag=$(date +%s)
$T 8 16 4096 128 1 --rounds 12 --only 8 > gpurun_out/r03z/c_$tag.log 2>&1
grep -h "per XCD (work\|last launch" gpurun_out/r03z/c_$tag.log | cut -c1-330
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2 || true
