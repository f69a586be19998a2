"""Wait timeline of a kernel from the compiled ISA: per s_waitcnt, the instructions issued since the previous one.
usage: python scratch/isa_timeline.py <mangled-name-substring> [--brief]"""
import subprocess, sys, re
S = "/tmp/lt_isa.s"
if "--nobuild" not in sys.argv:
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-I/root/repo/include", "--cuda-device-only", "-S",
                    "/root/repo/lap-time-optimization_amd/csrc/ltompc.hip", "-o", S], stderr=subprocess.DEVNULL, check=True)
src = open(S).read().split("\n")
key = sys.argv[1]
starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
beg = [i for i in starts if key in src[i]][0]
end = min([i for i in starts if i > beg] + [len(src)])
lines = [l.strip() for l in src[beg:end] if l.strip() and not l.strip().startswith((";", "."))]
tl, c = [], dict(valu=0, gl=0, gs=0, sl=0, scl=0, scs=0, lds=0)
tot = dict(c)
for l in lines:
    op = l.split()[0]
    if op.startswith("s_waitcnt"):
        tl.append((dict(c), l)); c = {k: 0 for k in c}
        continue
    k = ("valu" if op.startswith("v_") else "gl" if op.startswith("global_load") else "gs" if op.startswith("global_store") else
         "sl" if op.startswith("s_load") else "scl" if op.startswith("scratch_load") else "scs" if op.startswith("scratch_store") else
         "lds" if op.startswith("ds_") else None)
    if k: c[k] += 1; tot[k] += 1
print(src[beg][:100], "waits:", len(tl), "totals:", tot)
if "--brief" not in sys.argv:
    for c, l in tl:
        print("%5d valu %3d gload %3d gstore %3d sload %3d scr_ld %3d scr_st %3d lds | %s" % (c["valu"], c["gl"], c["gs"], c["sl"], c["scl"], c["scs"], c["lds"], l))
