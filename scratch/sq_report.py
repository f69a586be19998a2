"""Per-kernel SQ counter table (full-width launches only) from the passes of scratch/sq_passes.sh."""
import collections, csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sq"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        k = re.sub(r"[<(].*", "", r["Kernel_Name"].replace("void ", "").replace("ltompc::", "")).strip()
        vals[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
kern = ["k_eval", "k_expand", "k_riccati8", "k_linesearch", "k_update"]
names = sorted({c for k in kern for c in vals[k]})
print(f"{'counter':28s}" + "".join(f"{k:>16s}" for k in kern))
tab = {}
for c in names:
    row = []
    for k in kern:
        v = vals[k].get(c, [])
        if not v:
            row.append(float("nan")); continue
        g = max(x for x, _ in v)
        full = [y for x, y in v if x == g]
        row.append(sum(full) / len(full))
    tab[c] = row
    print(f"{c:28s}" + "".join(f"{x:16.4g}" for x in row))
print()
def ratio(a, b, label):
    if a not in tab or b not in tab:
        return
    print(f"{label:28s}" + "".join(f"{(x / y if y else float('nan')):16.3f}" for x, y in zip(tab[a], tab[b])))
ratio("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "wave cycles / VALU inst")
ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "VALU active / wave cycles")
ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wait inst any / wave cyc")
ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wait any / wave cycles")
ratio("SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", "VMEM active / wave cycles")
ratio("SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "scalar active / wave cyc")
ratio("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SALU / VALU insts")
ratio("SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "VMEM rd / VALU insts")
ratio("SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "VMEM wr / VALU insts")
ratio("SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU", "trans f64 / VALU")
ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "VMEM level / rd inst (lat)")
ratio("SQ_INSTS_VALU", "SQ_WAVES", "VALU insts per wave")
ratio("SQ_INSTS_VMEM_RD", "SQ_WAVES", "VMEM rd per wave")
ratio("SQ_INSTS_VMEM_WR", "SQ_WAVES", "VMEM wr per wave")
ratio("SQ_WAVE_CYCLES", "SQ_WAVES", "cycles per wave")
