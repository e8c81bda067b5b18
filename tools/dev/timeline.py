import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = [r for r in csv.DictReader(open(f)) if "mm_" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower() or "memset" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:60]}")
