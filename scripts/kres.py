"""Compact kernel resource table of one .hip file: python scripts/kres.py lv_match.hip [filter] (cross-compiles for gfx950 with the Makefile's flags)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
os.makedirs("/tmp/kres", exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "/tmp/kres/x.o",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(ROOT, "limo-velo_amd/csrc")).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+[A-Za-z])(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in name: continue
    print(f"{name[-60:]:60s} VGPR {v.get('VGPRs',-1):4d} spill {v.get('VGPRs Spill',-1):3d} scratch {v.get('ScratchSize',-1):5d} LDS {v.get('LDS Size',-1):7d} occ {v.get('Occupancy',-1)}")
