# Same-box A/B of the level-0 item-stream kernel's workgroup shape (AL3D_R16_SHAPE: 0 = one 12-wave workgroup per CU,
# 4 = three 4-wave workgroups per CU) in the pipelined and the serial bench.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion"
: > $O/ab_r16_shape.txt
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py $ARGS > $O/ab_knob.json 2> $O/ab_knob.err || return 1
  python3 -c "
import json,sys
d=json.loads(open('$O/ab_knob.json').read().strip().splitlines()[-1])
l=[round(r['avg_us']) for r in d['roofline_sparse']['layers'][:6]]
print('$name', d['value'], d['ms_per_step'], d['selected_equals_oracle'], 'sparse ms/batch', d['roofline_sparse']['ms_per_batch'], 'L0 us', l)" >> $O/ab_r16_shape.txt
}
run shape0 AL3D_R16_SHAPE=0 && run shape4 AL3D_R16_SHAPE=4 && run shape0 AL3D_R16_SHAPE=0 && run shape4 AL3D_R16_SHAPE=4 && run serial_shape0 AL3D_PIPELINE=0 AL3D_R16_SHAPE=0 && run serial_shape4 AL3D_PIPELINE=0 AL3D_R16_SHAPE=4
cat $O/ab_r16_shape.txt
