"""Condense the output of tools/bench_conv_split.py (one or more A/B sections) into a table."""
import re, sys
for l in open(sys.argv[1]):
    if l.startswith('==='):
        print(l.strip()); continue
    m = re.match(r'(\S+\s+\d+->\s*\d+\s+\d+x\s*\d+) \|', l)
    if m:
        parts = re.findall(r'(\w+):\s+([\d.]+) us\s+([\d.]+) TF', l)
        print(m.group(1), ' | '.join(f"{n} {us:>7s}us {tf:>6s}TF" for n, us, tf in parts))
    elif 'total' in l:
        print(l.strip())
