# Same-box A/B of the sweep's scheduling knobs at HEAD (bench.py headline only): default, index work after the sparse
# stage, + NMS after the sparse stage, high-priority main stream.  One process per setting, back to back.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion"
: > $O/ab_pipeline_knobs.txt
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py $ARGS > $O/ab_knob.json 2> $O/ab_knob.err || return 1
  python3 -c "
import json,sys
d=json.loads(open('$O/ab_knob.json').read().strip().splitlines()[-1])
print('$name', d['value'], d['ms_per_step'], d['selected_equals_oracle'])" >> $O/ab_pipeline_knobs.txt
}
run default A=1 && run side_after_sparse AL3D_SIDE_AFTER_SPARSE=1 && run side+nms_after_sparse AL3D_SIDE_AFTER_SPARSE=1 AL3D_NMS_AFTER_SPARSE=1 && run main_priority AL3D_MAIN_PRIORITY=1 && run side_after+priority AL3D_SIDE_AFTER_SPARSE=1 AL3D_MAIN_PRIORITY=1 && run default_again A=1
cat $O/ab_pipeline_knobs.txt
