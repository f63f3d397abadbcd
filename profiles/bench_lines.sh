set -e
mkdir -p gpurun_out/profiles_r03
for wl in cfg2 cfg2nc cfg1 cfg1c cfg3; do
  python3 bench.py --workload $wl > gpurun_out/profiles_r03/r03_bench_$wl.json 2> gpurun_out/profiles_r03/r03_bench_$wl.err
  tail -c 300 gpurun_out/profiles_r03/r03_bench_$wl.json
done
python3 bench.py --weights bf16 > gpurun_out/profiles_r03/r03_bench_cfg2_weights_bf16.json 2>/dev/null
