"""Dev tool: tabulate gpurun_out/r2_glds_abl.log (tools/glds_ablate.sh): mean us per layer shape and variant."""
import collections, re, sys
rows = collections.OrderedDict()
cur = None
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2_glds_abl.log"):
    if l.startswith("==="):
        cur = l.strip()[4:]
        continue
    m = re.search(r"(\d+)->\s*(\d+) K=\s*(\d+).*?:\s+([\d.]+) us", l)
    if m:
        rows.setdefault(m.group(1, 2, 3), collections.OrderedDict()).setdefault(cur, []).append(float(m.group(4)))
hdr = None
tot = collections.OrderedDict()
for k, v in rows.items():
    if hdr is None:
        hdr = list(v.keys())
        print("layer".ljust(14), " ".join(h.replace("ABL=", "A").replace(" CFG=", "C").rjust(8) for h in hdr))
    print(("%s->%s K%s" % k).ljust(14), " ".join(("%.0f" % (sum(v[h]) / len(v[h]))).rjust(8) if h in v else "     n/a" for h in hdr))
    for h in hdr:
        tot[h] = tot.get(h, 0.0) + sum(v.get(h, [0.0]))
print("total/batch".ljust(14), " ".join(("%.0f" % tot[h]).rjust(8) for h in hdr))
