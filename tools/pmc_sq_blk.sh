# Dev tool (GPU box): SQ counters of the block-staged sparse kernels against the kernels they replace
# (tools/bench_blk.py, batch 32), two passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bka /tmp/p_bkb
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p_bka -o p -- python3 $R/tools/bench_blk.py 32 2 ${BLK_MODES:-off,on} > $O/pmc_bka.log 2> $O/pmc_bka.err || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace -d /tmp/p_bkb -o p -- python3 $R/tools/bench_blk.py 32 2 ${BLK_MODES:-off,on} > $O/pmc_bkb.log 2> $O/pmc_bkb.err || exit 1
python3 - <<PY > $O/pmc_blk.txt
import sqlite3, glob, re
acc={}
for d in ("/tmp/p_bka","/tmp/p_bkb"):
    db=glob.glob(d+"/**/*.db", recursive=True)[0]
    c=sqlite3.connect(db)
    for k,cn,n,s,dur in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "sp_conv" in k and re.search(r"<(32, 32|64, 64|128, 128)", k):
            a=acc.setdefault(k[:60],{}); a[cn]=s/n; a["dur_us"]=dur/1e3; a["n"]=n
for k,v in sorted(acc.items()):
    print(k)
    print("   ", "  ".join(f"{cn.replace('SQ_','')}={val/1e6:.2f}M" if cn not in ("dur_us","n") else f"{cn}={val:.1f}" for cn,val in sorted(v.items())))
PY
cat $O/pmc_blk.txt
