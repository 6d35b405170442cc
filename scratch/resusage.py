"""kernel resource usage from a hipcc -Rpass-analysis=kernel-resource-usage log: python scratch/resusage.py <log> [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read(); pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
rows = []
for b in blocks:
    name = b.split('\n')[0].strip().split()[0]
    g = lambda k: (lambda m: int(m.group(1)) if m else -1)(re.search(k + r': (\d+)', b))
    rows.append((name, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill'), g('VGPRs Spill')))
dm = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, d in zip(rows, dm):
    d = d.replace('mer::', '').replace('void ', '')
    if pat in d:
        print(d[:70].ljust(70), 'vgpr %d agpr %d scratch %d occ %d sspill %d vspill %d' % r[1:])
