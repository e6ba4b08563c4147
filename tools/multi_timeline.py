"""Window of a rocprofv3 kernel trace of bench.py's timed loop: the launches that serve the batches (k_sweep_multi),
microseconds from the window's first dispatch.  usage: python tools/multi_timeline.py <kernel_trace.csv> > profiles/round2_multi_timeline.txt"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: (re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"]) or [r["Kernel_Name"][:30]])[0]
multi = [i for i, r in enumerate(rows) if "k_sweep_multi" in r["Kernel_Name"] or "k_sweep_lean_multi" in r["Kernel_Name"]]
lo = multi[len(multi) - 60]
t0 = int(rows[lo]["Start_Timestamp"])
print("# rocprofv3 --kernel-trace of `python3 bench.py --headline-only --steps 200` (tools/gpu_f.sh): a window of the timed loop,")
print("# microseconds from the window's first dispatch.  Two batches of 32 queries alternate on two streams; each line is ONE")
print("# launch serving 32 queries (a step enqueues one batch and fetches the previous one's 32 results): the launches follow")
print("# one another without a gap - the step is bound by the kernel, not by the host.")
print("#    start       end     dur  queue  kernel")
for r in rows[lo:lo + 24]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f'{s:10.2f} {e:9.2f} {e - s:7.2f}  q{r.get("Queue_Id", "?"):>3}  {name(r)}')
