# Dev tool (GPU box): SQ counters of the Swin-T token kernels (wave cycles, parked / issue-stalled / active shares, MFMA busy
# cycles, VALU instructions) -- bash tools/pmc_sq_swin.sh under gpurun; prints per-kernel averages.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_pm
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d /tmp/p_pm -o p -- python3 $R/tools/bench_swin.py 16 1 > $O/pmc_mlp.log 2> $O/pmc_mlp.err
DB=$(find /tmp/p_pm -name "*.db" | head -1)
python3 - <<PY
import sqlite3
c=sqlite3.connect("$DB")
rows=c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
acc={}
for k,cn,n,s in rows:
    if "tok_mlp" in k or "tok_attn_block" in k or "tok_window" in k:
        acc.setdefault(k[:45],{})[cn]=s/n
for k,v in acc.items():
    w=v.get("SQ_WAVE_CYCLES",1)
    print(k)
    for cn,val in sorted(v.items()): print("    %-28s %14.0f  %5.1f%% of wave cycles" % (cn, val, 100*val/w))
PY
