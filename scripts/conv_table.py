import re,sys
for l in sys.stdin:
    m=re.match(r'(.{24}) fwd\s+([\d.]+) us.*?dgrad\s+([\d.]+) us.*?wgrad\s+([\d.]+) us.*?torch fwd\s+([\d.]+) us.*?dgrad\s+([\d.]+) us wgrad\s+([\d.]+) us',l)
    if m:
        n,f1,d1,w1,f2,d2,w2=m.groups()
        print(f"{n} fwd {f1:>7} | {f2:>7}   dgrad {d1:>7} | {d2:>7}   wgrad {w1:>7} | {w2:>7}")
    elif l.startswith("B=") or l.startswith("mode"): print(l.strip())
