# Dev tool (GPU box): SQ counters of the dense kernels (tools/bench_conv.py 64 f16x3dma), both wave shapes of the 3x3 kernel.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_dense.txt
for shape in 0 1; do
  rm -rf /tmp/p_da /tmp/p_db
  AL3D_FRAG_SHAPE=$shape timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p_da -o p -- python3 $R/tools/bench_conv.py 64 f16x3dma > $O/pmc_da.log 2> $O/pmc_da.err || exit 1
  AL3D_FRAG_SHAPE=$shape timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace -d /tmp/p_db -o p -- python3 $R/tools/bench_conv.py 64 f16x3dma > $O/pmc_db.log 2> $O/pmc_db.err || exit 1
  python3 - <<PY >> $O/pmc_dense.txt
import sqlite3, glob
acc={}
for d in ("/tmp/p_da","/tmp/p_db"):
    db=glob.glob(d+"/**/*.db", recursive=True)[0]
    c=sqlite3.connect(db)
    for k,cn,n,s,dur in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "conv3x3_f16x3_frag" in k or "dma2" in k:
            a=acc.setdefault(k[:60],{}); a[cn]=s/n; a["dur_us"]=dur/1e3; a["n"]=n
print("== AL3D_FRAG_SHAPE=$shape")
for k,v in sorted(acc.items()):
    print(k)
    print("   ", "  ".join(f"{cn.replace('SQ_','')}={val/1e6:.2f}M" if cn not in ("dur_us","n") else f"{cn}={val:.1f}" for cn,val in sorted(v.items())))
PY
done
cat $O/pmc_dense.txt
