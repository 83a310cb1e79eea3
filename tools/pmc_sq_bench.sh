# Dev tool (GPU box): the same SQ counters for the headline bench's kernels (serial mode, 16 scenes).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_pb
export AL3D_PIPELINE=0
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d /tmp/p_pb -o p -- python3 $R/bench.py --scenes 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-math --no-from-files --no-verify --no-bevfusion > $O/pmc_bench.log 2> $O/pmc_bench.err
DB=$(find /tmp/p_pb -name "*.db" | head -1)
python3 - <<PY
import sqlite3
c=sqlite3.connect("$DB")
rows=c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
acc={}
for k,cn,n,s in rows:
    if any(t in k for t in ("sp_conv","conv3x3_f16x3_frag","conv2d_f16x3_dma2","head_nms","sp_table_rows27","vox_first")):
        acc.setdefault(k[:60],{})[cn]=(s/n,n)
for k,v in sorted(acc.items()):
    w=v.get("SQ_WAVE_CYCLES",(1,0))[0]
    print(k, "launches", v["SQ_WAVE_CYCLES"][1])
    print("    " + "  ".join("%s %.1f%%" % (cn.replace("SQ_",""), 100*val/w) for cn,(val,_) in sorted(v.items()) if cn!="SQ_WAVE_CYCLES"), " wave_quadcycles %.0fM" % (w/1e6))
PY
