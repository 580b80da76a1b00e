set -x
mkdir -p gpurun_out/pmc
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k, v['avg_ms'], v['frac']) for k,v in d['kernels'].items()]"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|GRBM_GUI_ACTIVE|SQ_INSTS_VALU\b|SQ_ACTIVE_INST_VALU|SQ_INST_CYCLES_VMEM|SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK_CONFLICT" | sort -u | tr '\n' ' '
echo
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc -o sq --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/sq.log 2>&1
ls gpurun_out/pmc
