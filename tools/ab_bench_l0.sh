# dev: the headline bench under the level-0 row orders (same box, back to back)
cd $GRAFT_REPO_ROOT
O=gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra-math --no-from-files --no-bevfusion"
for cfg in ${CFGS:-"off,16" "raster,16" "off,16" "raster,16"}; do
  IFS=, read l0 c1 c2 <<< "$cfg"
  AL3D_L0=$l0 AL3D_R16_COUTS=$c1${c2:+,$c2} AL3D_R16_TPW=${TPW:-8} timeout -k 10 300 python bench.py $ARGS > $O/ab_l0_$l0$c2.json 2> $O/ab_l0_$l0$c2.err || { tail -5 $O/ab_l0_$l0$c2.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/ab_l0_$l0$c2.json"))
rs=d.get("roofline_sparse",{})
print("$cfg", "frames/s", d["value"], "ms/step", d["ms_per_step"], "sparse ms/batch", rs.get("ms_per_batch"), "equal_oracle", d.get("selected_equals_oracle"), [round(l["avg_us"]) for l in rs.get("layers",[])][:7])
PY
done
