"""Summary of tools/fetch_calib.sh: per calibration kernel, what rocprofv3's FETCH_SIZE (x 1024 B) reports against the bytes the kernel is
known to read, next to the TCC / TCP request counters of the same kernel. The last column is the factor to multiply FETCH_SIZE x 1024 with
to get the requested bytes of that access shape (the guide's x2 holds for wide coalesced streaming reads only)."""
import collections, csv, glob, re, sys
root = sys.argv[1]
known, ms = {}, {}
for line in open(f"{root}/plain.txt"):
    m = re.match(r"(\S+)\s+known_bytes\s+(\d+)\s+([\d.]+) ms", line)
    if m: known[m.group(1)] = float(m.group(2)); ms[m.group(1)] = float(m.group(3))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if name == "calib_gather_64B_node_resident<0>": continue
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
print("FETCH_SIZE calibration, MI355X (gfx950), rocprofv3: counters per kernel (one dispatch each)")
print(f"{'kernel':42s} {'known bytes':>14s} {'FETCH_SIZE*1024':>16s} {'reported/known':>14s} {'factor':>8s}   other counters")
for k in known:
    c = tot.get(k, {})
    fs = c.get("FETCH_SIZE", float('nan')) * 1024.0
    ratio = fs / known[k] if known[k] else float('nan')
    other = "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items()) if n != "FETCH_SIZE")
    print(f"{k:42s} {known[k]:14.0f} {fs:16.0f} {ratio:14.4f} {1.0 / ratio if ratio else float('nan'):8.3f}   {other}")
print()
print("plain run (HIP events):")
print(open(f"{root}/plain.txt").read())
