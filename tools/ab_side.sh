# dev: placement / CU share of the side streams' work with the round-4 level-0 kernels (same box, back to back)
# CFGS="side_after_sparse,nms_after_sparse,nms_cus ..."
cd $GRAFT_REPO_ROOT
O=gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion"
for cfg in ${CFGS:-"0,0,0" "1,0,0" "1,1,0" "0,0,0" "1,1,0"}; do
  IFS=, read sas nas nms <<< "$cfg"
  AL3D_SIDE_AFTER_SPARSE=$sas AL3D_NMS_AFTER_SPARSE=$nas AL3D_NMS_CUS=$nms timeout -k 10 300 python bench.py $ARGS > $O/ab_side.json 2> $O/ab_side.err || { tail -5 $O/ab_side.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/ab_side.json"))
rs=d.get("roofline_sparse",{})
print("side_after_sparse=$sas nms_after_sparse=$nas nms_cus=$nms", "frames/s", d["value"], "sparse ms/batch", rs.get("ms_per_batch"), "dense avg us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], [round(l["avg_us"]) for l in rs.get("layers",[])][:6])
PY
done
