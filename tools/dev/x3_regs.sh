#!/bin/bash
# register / spill table of every conv3d_x3_kernel instantiation (compile-only, no GPU needed)
hipcc --offload-arch=gfx950 -O3 -c rc_mvsnet_amd/csrc/conv3d_x3.hip -o /tmp/x3_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
name = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: name = m.group(1); vals = {}
    m = re.search(r"remark: +(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and name: vals[m.group(1)] = int(m.group(2))
    if "LDS Size" in line and name:
        if "conv3d_x3_kernel" in name:
            t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
            print("x3<%s,%s,kind %s,NP %s>  VGPRs %3d  AGPRs %3d  spill %3d  scratch %4d" % (*t.groups(), vals.get("VGPRs",0), vals.get("AGPRs",0), vals.get("VGPRs Spill",0), vals.get("ScratchSize [bytes/lane]",0)))
        name = None
'
