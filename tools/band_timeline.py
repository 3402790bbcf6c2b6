"""Timeline of the last frame in a rocprofv3 kernel trace of tools/band_trace.py: per launch start offset / duration (us)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rrtd::", "")
# a frame starts at its k_sample_f32 / k_raygen launch
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_sample_f32") or name(r).startswith("k_raygen<")]
rows = rows[starts[-int(sys.argv[2]) if len(sys.argv) > 2 else -1]:]
t0 = int(rows[0]["Start_Timestamp"])
end_prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - end_prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name(r)[:60]}")
    end_prev = max(end_prev, e)
print("frame", (end_prev - t0) / 1e3, "us")
