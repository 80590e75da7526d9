#!/usr/bin/env python3
"""tools/kres.py [file.hip] -- register / scratch / occupancy of every kernel instantiation (the compiler's view; no GPU needed)"""
import os
import re
import subprocess
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mimc3_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 else "match_px_kernel.hip"
p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", "-c",
                    os.path.join(here, src), "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = {}
for line in p.stderr.splitlines():
    t = line.strip()
    if "Function Name:" in t:
        cur = {"name": t.split("Function Name:")[1].strip()}
    for key, tag in (("VGPRs:", "vgpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("SGPRs:", "sgpr")):
        if key in t and "name" in cur and "AGPR" not in t.split(key)[0][-3:]:
            cur.setdefault(tag, t.split(":")[-1].split("[")[0].strip())
    if "LDS Size" in t and "name" in cur:
        d = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        d = d.replace("mimc3::", "").replace("void ", "")
        print("%-88s vgpr %4s sgpr %4s scratch %10s occ %s" % (d[:88], cur.get("vgpr"), cur.get("sgpr"), cur.get("scratch"), cur.get("occ")))
        cur = {}
