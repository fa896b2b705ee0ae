cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/trace3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/trace3_bench.log 2>&1
find gpurun_out/trace3 -name "*kernel_trace.csv" | head
