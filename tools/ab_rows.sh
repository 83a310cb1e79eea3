cd $GRAFT_REPO_ROOT
O=gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-from-files --no-bevfusion"
for cfg in pair f32 pair f32; do
  AL3D_L0_ROWS=$cfg timeout -k 10 300 python bench.py $ARGS > $O/ab_rows.json 2> $O/ab_rows.err || { tail -5 $O/ab_rows.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/ab_rows.json"))
rs=d.get("roofline_sparse",{})
print("l0_rows=$cfg", "frames/s", d["value"], "sparse ms/batch", rs.get("ms_per_batch"), [round(l["avg_us"]) for l in rs.get("layers",[])][:6], d.get("selection_f16x3_vs_bf16x6"))
PY
done
