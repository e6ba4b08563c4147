"""Dev helper: summarise a rocprofv3 kernel trace CSV — per kernel name, start/end relative to the first, for a
window of consecutive dispatches in steady state (shows overlap between streams)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + n]:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.2f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.2f}  q{r.get("Queue_Id", "?"):>3}  {name}')
