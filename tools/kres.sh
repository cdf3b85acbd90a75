#!/bin/bash
# kernel resource usage of one csrc file: name, VGPRs, spilled VGPRs, LDS bytes.  usage: tools/kres.sh xgemm2 [name filter] [extra hipcc flags]
f=$1; pat=${2:-.}; shift; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/makani_amd/csrc -I/root/repo/include "$@" \
  -c /root/repo/makani_amd/csrc/$f.hip -o /tmp/kres_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import sys, re
name = None; d = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: name = m.group(1); d = {}
    for key in ("VGPRs", "VGPRs Spill", "LDS Size \\[bytes/block\\]", "ScratchSize \\[bytes/lane\\]"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m: d[key] = m.group(1)
    if "LDS Size" in line and name:
        print(name, "vgpr", d.get("VGPRs"), "spill", d.get("VGPRs Spill"), "scratch", d.get("ScratchSize \\[bytes/lane\\]"), "lds", d.get("LDS Size \\[bytes/block\\]"))
' | grep -E "$pat"
