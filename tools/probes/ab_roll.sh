set -e
mkdir -p gpurun_out/r4d
python3 tools/ab_knob.py --knob apply_variant --values 56 63 --rounds 4 > gpurun_out/r4d/ab_roll_116.txt 2>&1
cat gpurun_out/r4d/ab_roll_116.txt
python3 -m pytest tests/test_gpu_parity.py -x -q -k "lattice or full_size_bench" 2>&1 | tail -3
