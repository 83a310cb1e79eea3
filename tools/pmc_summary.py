"""Print per-kernel averages of every counter in a rocprofv3 --pmc rocpd database (values / 1e6)."""
import re, sqlite3, sys
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection "
                     "where kernel_name like ? group by kernel_name,counter_name", (f"%{sys.argv[1]}%",)).fetchall()
    d = {}
    for k, cn, v, n, dur in rows:
        key = re.sub(r"\(.*", "", re.sub(r"^void ", "", k))
        d.setdefault(key, {})[cn.replace("SQ_", "")] = round(v / 1e6, 2)
        d[key]["dur_us"] = round(dur / 1e3, 1)
    for k, v in d.items():
        print(k, v)
