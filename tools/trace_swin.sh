# Dev tool (GPU box): per-dispatch durations of one Swin-T forward at 16 samples (gpurun_out/swin16_seq.csv).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_sw
rocprofv3 --kernel-trace -d /tmp/p_sw -o s -- python3 $R/tools/bench_swin.py 16 1 > $O/swin16.log 2> $O/swin16.err
DB=$(find /tmp/p_sw -name "*.db" | head -1)
python3 - <<PY
import sqlite3
c=sqlite3.connect("$DB")
print([r[1] for r in c.execute("pragma table_info(kernels)").fetchall()])
PY
python3 $R/tools/rocpd_summary.py seq $DB $O/swin16_seq.csv 110
tail -2 $O/swin16.log
