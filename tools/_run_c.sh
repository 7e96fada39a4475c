set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r2_c_trace -o t --output-format csv -- python3 $R/tools/halo_overhead_self.py --iters 20 > $R/gpurun_out/r2_c_halo.txt 2>&1
cd $R
f=$(ls gpurun_out/r2_c_trace/*/t_kernel_trace.csv gpurun_out/r2_c_trace/t_kernel_trace.csv 2>/dev/null | head -1)
echo $f; wc -l $f
python tools/trace_gaps.py $f cgm_update_kernel 400 > gpurun_out/r2_c_gaps.txt
tail -6 gpurun_out/r2_c_halo.txt
cp $f gpurun_out/r2_c_kernel_trace.csv; rm -rf gpurun_out/r2_c_trace
