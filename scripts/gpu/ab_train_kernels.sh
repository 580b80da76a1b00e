# A/B of the training kernels on ONE box, interleaved: LIBS="base new <variant> ..." -- "new" is the
# in-tree build, anything else scripts/probes/variants/libffn_<name>.so (MODES=bf16x6 MODEL=tiny by
# default); the rows carry a digest of the gradients (knock-out variants compute wrong ones)
MODES=${MODES:-bf16x6}
MODEL=${MODEL:-tiny}
RAYS=${RAYS:-65536}
LIBS=${LIBS:-base new}
REPS=${REPS:-3}
mkdir -p gpurun_out/ab
for rep in $(seq 1 $REPS); do
  for lib in $LIBS; do
    if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
    echo "$lib $rep $(python scripts/microbench_train_kernels.py --iters 6 --modes $MODES --model $MODEL --rays $RAYS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({m: d[m] for m in '$MODES'.split(',')})")" | tee -a gpurun_out/ab/train_kernels.txt
  done
done
