import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
start = int(sys.argv[2]); n = int(sys.argv[3])
prev_end = None
for r in rows[start:start + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")[:58]
    print(f"gap {gap:6.1f} us  dur {(e-s)/1e3:7.1f} us  grid=({int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {name}")
