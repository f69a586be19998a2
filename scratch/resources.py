"""Print VGPR / scratch / spill / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-I/root/repo/include", "--cuda-device-only",
       "-c", "/root/repo/lap-time-optimization_amd/csrc/ltompc.hip", "-o", "/tmp/lt_res.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0].strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':28s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'vspill':>7s} {'sspill':>7s} {'occ':>4s} {'LDS':>6s}")
for k, r in rows.items():
    print(f"{k:28s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('ScratchSize',0):8d} {r.get('VGPRs Spill',0):7d} {r.get('SGPRs Spill',0):7d} {r.get('Occupancy',0):4d} {r.get('LDS Size',0):6d}")
