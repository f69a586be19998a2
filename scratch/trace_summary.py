import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    if len(v) < 20: continue
    # second make_step = last half; print launches 0..24 of the last solve
    n = len(v); half = v[n - 61:] if n > 61 else v
    print(f"{k:40s} n={n} last-solve first 12 (us): " + " ".join(f"{x:.0f}" for x in half[:12]) + "  | 20..25: " + " ".join(f"{x:.0f}" for x in half[20:26]) + "  | 40..45: " + " ".join(f"{x:.0f}" for x in half[40:46]))
print("VGPR/LDS:", {r["Kernel_Name"].split("(")[0]: (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")) for r in rows})
