mkdir -p gpurun_out/r4aa
for rep in 1 2; do
  for v in 0 1; do
    BP5_GHOST_COMBINE_ON_COMM=$v python3 tools/halo_overhead_self.py --cell-block 4 4 2 --modes auto,none --iters 100 > gpurun_out/r4aa/halo_knob${v}_$rep.txt 2>&1
    echo "ghost_combine_on_comm=$v rep $rep: $(grep -E 'automatic: [0-9.]+ ms|no exchange: [0-9.]+ ms' gpurun_out/r4aa/halo_knob${v}_$rep.txt | cut -c1-90 | tr '\n' '|')"
  done
done
timeout -k 10 800 python -m pytest tests/test_gpu_multirank_loopback.py -x -q -k "undivided" > gpurun_out/r4aa/loopback.log 2>&1; tail -3 gpurun_out/r4aa/loopback.log
