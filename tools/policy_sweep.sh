# Dev tool: same-box A/B of sparse-kernel policies (AL3D_GLDS_PAIRS) and batch sizes; prints frames/s
for v in "32x32,64x64 64" "32x32,64x64,16x16,16x32 64" "32x32,64x64,128x128 64" "32x32,64x64,32x64,64x128 64" "32x32,64x64 96" "32x32,64x64 128" "32x32,64x64 64"; do
  set -- $v
  AL3D_GLDS_PAIRS=$1 timeout -k 10 300 python bench.py --batch $2 --no-cpu-baseline --no-extra-math --no-from-files > gpurun_out/pol.json 2> gpurun_out/pol.err || tail -3 gpurun_out/pol.err
  python -c "
import json;d=json.load(open('gpurun_out/pol.json'));print('pairs=$1 batch=$2', d['value'], d['roofline_sparse']['ms_per_batch'], d['roofline']['achieved'], d['selected_equals_oracle'])"
done
