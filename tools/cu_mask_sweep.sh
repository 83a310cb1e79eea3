# Dev tool: same-box A/B of CU masks for the side streams (AL3D_SIDE_CUS / AL3D_NMS_CUS); prints frames/s
for v in "0 0" "64 0" "32 0" "16 0" "0 16" "32 16" "64 32" "0 0"; do
  set -- $v
  AL3D_SIDE_CUS=$1 AL3D_NMS_CUS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-math > gpurun_out/cu_$1_$2.json 2> gpurun_out/cu_$1_$2.err || { tail -3 gpurun_out/cu_$1_$2.err; }
  python -c "
import json;d=json.load(open('gpurun_out/cu_$1_$2.json'));print('side_cus=$1 nms_cus=$2', d['value'], d['roofline_sparse']['ms_per_batch'], d['roofline']['achieved'], d['selected_equals_oracle'])"
done
