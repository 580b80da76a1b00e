# HBM traffic of the MLP kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limit),
# kernel-trace only (no other trace domains alongside --pmc).
mkdir -p gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/traffic -o fetch --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-render --no-target-shape > gpurun_out/traffic/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/traffic -o write --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-render --no-target-shape > gpurun_out/traffic/write.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/traffic -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-render --no-target-shape > gpurun_out/traffic/stats.log 2>&1
tail -1 gpurun_out/traffic/stats.log | cut -c1-300
ls gpurun_out/traffic
