# Dev tool (GPU box): FETCH_SIZE calibration by load shape (tools/probes/fetch_calib.hip) -> gpurun_out/fetch_calib.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_fc
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fc -o p -- $R/tools/probes/fetch_calib > $O/fetch_calib.log 2> $O/fetch_calib.err || exit 1
python3 - <<PY > $O/fetch_calib.json
import sqlite3, glob, json, re
db = glob.glob("/tmp/p_fc/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
actual = 2 * 1024**3
res = {}
for k, n, s, dur in c.execute("select kernel_name, count(*), sum(value), avg(duration) from counters_collection "
                              "where counter_name = 'FETCH_SIZE' group by kernel_name"):
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", k))
    counted = s / n * 1024.0                                   # FETCH_SIZE is in KB
    used = actual * (12 / 64 if name.startswith("x3_sector") else 1.0)
    res[name] = {"launches": n, "counted_mb": round(counted / 1e6, 1), "actual_mb": round(used / 1e6, 1),
                 "counted_over_actual": round(counted / used, 4), "avg_us": round(dur / 1e3, 1),
                 "gbs": round(used / dur, 1)}
print(json.dumps(res, indent=1))
PY
cat $O/fetch_calib.json
