"""Turn rocprofv3 rocpd (.db) outputs into the summaries committed under profiles/.

  python tools/rocpd_summary.py stats  <kernel-trace.db> <out.csv>
  python tools/rocpd_summary.py hbm    <FETCH_SIZE.db> <WRITE_SIZE.db> <out.json> [note]

``stats`` reproduces rocprofv3 --stats' kernel table (calls, total/avg/min/max ns, share).
``hbm`` sums the two single-counter passes per kernel and applies the gfx950 correction from
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE (KB) under-counts 128-byte requests as 64 B, so it
is doubled; WRITE_SIZE (KB) is taken as is.  Values are averages per launch.
"""
import csv
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, calls, tot, avg, mn, mx in rows:
            w.writerow([n, calls, tot, round(avg, 3), round(100.0 * tot / total, 2), mn, mx])


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    acc = {}
    for name, n, s in c.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                "where counter_name = ? group by kernel_name", (counter,)):
        k = short(name)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += n
        a[1] += s
    return acc


# Kernels whose global loads fetch 64-byte segments (16 f32 channels of a pixel / of a 16-channel row): FETCH_SIZE
# tallies those correctly.  Everything else issues fully coalesced 16 B/lane loads or LDS-DMA, whose 128-byte
# requests gfx950 tallies at 64 B (MI355X_MICROARCH.md): FETCH_SIZE x 2.
NO_X2 = ("conv3x3_f16x3", "conv2d_f16x3_kernel", "conv2d_f16x3_bstream", "sp_conv_wave2_kernel<16,")


def hbm(fdb, wdb, out, note):
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    res = {}
    for k, (n, kb) in fetch.items():
        wn, wkb = write.get(k, (0, 0.0))
        f_avg = kb / n
        w_avg = wkb / wn if wn else 0.0
        x2 = not k.startswith(NO_X2)
        res[k] = {"launches": n, "fetch_kb_raw": f_avg, "write_kb": w_avg,
                  "fetch_mb_raw": f_avg * 1024 / 1e6, "fetch_mb_x2": 2.0 * f_avg * 1024 / 1e6,
                  "write_mb": w_avg * 1024 / 1e6,
                  "hbm_mb_corrected": ((2.0 if x2 else 1.0) * f_avg + w_avg) * 1024 / 1e6,
                  "x2_applies": x2,
                  "correction": ("FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 tallies the 128-byte requests of fully "
                                 "coalesced 16 B/lane loads and LDS-DMA at 64 B); WRITE_SIZE as counted") if x2 else
                                ("none: this kernel's loads fetch 64-byte segments (16 f32 channels per pixel / row unit), "
                                 "which FETCH_SIZE tallies correctly -- raw FETCH_SIZE + WRITE_SIZE equals the algorithmic "
                                 "bytes of the launch mix"),
                  "note": note}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_mb_corrected"] * kv[1]["launches"]))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        note = sys.argv[5] if len(sys.argv) > 5 else ""
        hbm(sys.argv[2], sys.argv[3], sys.argv[4], note)
