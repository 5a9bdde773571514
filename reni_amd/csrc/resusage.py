"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks read from stdin."""
import re
import sys

cur = None
rows = []
keys = ['VGPRs:', 'AGPRs:', 'ScratchSize [bytes/lane]:', 'Occupancy [waves/SIMD]:', 'SGPRs:']
for line in sys.stdin:
    if ' error' in line or 'warning:' in line:
        print(line.rstrip())
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'n': m.group(1)}
        rows.append(cur)
        continue
    for k in keys:
        if cur is not None and ('    ' + k) in line:
            cur[k] = line.split(k)[1].split('[')[0].strip()
for r in rows:
    print(r['n'][:64].ljust(64), 'V', r.get(keys[0]), 'A', r.get(keys[1]), 'scr', r.get(keys[2]), 'occ', r.get(keys[3]), 'sg', r.get(keys[4]))
