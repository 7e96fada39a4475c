set -e
mkdir -p gpurun_out/r4b
python3 tools/ab_knob.py --knob update_flat,update_unroll,update_nt --values 0,4,0 1,1,0 1,2,0 1,4,0 1,1,1 1,2,1 0,4,1 --rounds 4 > gpurun_out/r4b/ab_update_116.txt 2>&1
cat gpurun_out/r4b/ab_update_116.txt
python3 tools/ab_knob.py --knob update_flat,update_unroll,update_nt --values 0,4,-1 1,1,-1 1,2,-1 1,1,0 --cells 54 54 54 --cell-block 4 4 2 --rounds 6 --iters 100 > gpurun_out/r4b/ab_update_54.txt 2>&1
cat gpurun_out/r4b/ab_update_54.txt
