# Dev tool (GPU box): MFMA instruction counts and wave cycles of the sparse kernels of levels 2-3 with their rows in raster order
# (AL3D_MASK_SORT="") and grouped by tap mask (default): serial mode, 16 scenes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export AL3D_PIPELINE=0
: > $O/pmc_sq_masksort.txt
for m in raster sorted; do
  rm -rf /tmp/p_pm
  if [ $m = raster ]; then export AL3D_MASK_SORT=""; else unset AL3D_MASK_SORT; fi
  timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d /tmp/p_pm -o p -- python3 $R/bench.py --scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion > $O/pmc_ms.log 2> $O/pmc_ms.err || exit 1
  python3 - <<PY >> $O/pmc_sq_masksort.txt
import sqlite3, glob
db=glob.glob("/tmp/p_pm/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
acc={}
for k,cn,n,s,dur in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
    if "sp_conv_wave2_kernel<128, 128" in k or "sp_conv_glds_kernel<64, 64" in k or "sp_conv_wave2_kernel<64, 128" in k or "sp_conv_wave2_kernel<32, 64" in k or "sp_conv_rng" in k:
        a=acc.setdefault(k[:48],{}); a[cn]=s/n; a["dur_us"]=dur/1e3; a["n"]=n
print("== levels 2-3 in $m order (per launch)")
for k,v in sorted(acc.items()):
    print("  %-48s n=%3d %8.1f us  MFMA insts %7.2fM  MFMA busy %8.1fM  wave quad-cycles %8.1fM  VMEM rd %6.2fM  LDS %6.2fM" % (k, v["n"], v["dur_us"], v.get("SQ_INSTS_MFMA",0)/1e6, v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1e6, v.get("SQ_WAVE_CYCLES",0)/1e6, v.get("SQ_INSTS_VMEM_RD",0)/1e6, v.get("SQ_INSTS_LDS",0)/1e6))
PY
done
cat $O/pmc_sq_masksort.txt
