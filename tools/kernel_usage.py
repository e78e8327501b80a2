#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (make -C myslam_amd/csrc asm)."""
import glob, re, sys
for f in sorted(glob.glob('myslam_amd/csrc/build/asm/*.usage.txt')):
    cur = {}
    for line in open(f):
        m = re.search(r'remark: +([\w \[\]/]+?): +(\S+) \[-Rpass', line)
        if not m: continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ('Function Name', 'Name'):
            if cur: print(cur)
            cur = {'k': v[:60]}
        elif k in ('VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]', 'TotalSGPRs', 'SGPRs'):
            cur[k.split(' ')[0]] = v
    if cur: print(cur)
