# Dev tool (GPU box): SQ counters of the level-0 kernels (tools/bench_l0.py, batch 32), two passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_l0a /tmp/p_l0b
export BENCH_L0_MODES=${BENCH_L0_MODES:-off,raster16+32p}
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p_l0a -o p -- python3 $R/tools/bench_l0.py 32 2 > $O/pmc_l0a.log 2> $O/pmc_l0a.err || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace -d /tmp/p_l0b -o p -- python3 $R/tools/bench_l0.py 32 2 > $O/pmc_l0b.log 2> $O/pmc_l0b.err || exit 1
python3 - <<PY > $O/pmc_l0.txt
import sqlite3, glob
acc={}
for d in ("/tmp/p_l0a","/tmp/p_l0b"):
    db=glob.glob(d+"/**/*.db", recursive=True)[0]
    c=sqlite3.connect(db)
    for k,cn,n,s,dur in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "sp_conv" in k and ("16, 16" in k or "16, 32" in k or "r16" in k):
            a=acc.setdefault(k[:70],{}); a[cn]=s/n; a["dur_us"]=dur/1e3; a["n"]=n
for k,v in sorted(acc.items()):
    print(k)
    print("   ", "  ".join(f"{cn.replace('SQ_','')}={val/1e6:.2f}M" if cn not in ("dur_us","n") else f"{cn}={val:.1f}" for cn,val in sorted(v.items())))
PY
cat $O/pmc_l0.txt
