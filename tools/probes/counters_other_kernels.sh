mkdir -p gpurun_out/r4f
BENCH_ARGS="--degree 8 --cells 46 46 46" bash tools/profile_counter_groups_short.sh r4f/p8_atomic > gpurun_out/r4f/p8_atomic.log 2>&1
BENCH_ARGS="--degree 8 --cells 46 46 46 --cell-block 2 2 2 --apply-variant 56" bash tools/profile_counter_groups_short.sh r4f/p8_block > gpurun_out/r4f/p8_block.log 2>&1
BENCH_ARGS="--degree 5 --cells 73 73 73" bash tools/profile_counter_groups_short.sh r4f/p5_block > gpurun_out/r4f/p5_block.log 2>&1
BENCH_ARGS="--operator helmholtz --degree 3 --cells 122 122 122" bash tools/profile_counter_groups_short.sh r4f/helmholtz_p3 > gpurun_out/r4f/helmholtz_p3.log 2>&1
BENCH_ARGS="--degree 1 --cells 367 367 367" bash tools/profile_counter_groups_short.sh r4f/p1_gauss > gpurun_out/r4f/p1_gauss.log 2>&1
BENCH_ARGS="--degree 1 --cells 367 367 367 --quadrature gll" bash tools/profile_counter_groups_short.sh r4f/p1_gll > gpurun_out/r4f/p1_gll.log 2>&1
ls gpurun_out/r4f
