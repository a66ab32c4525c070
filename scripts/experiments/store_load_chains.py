#!/usr/bin/env python3
"""Static audit of the device code for serialized read-modify-write chains: on gfx9-family parts stores count in vmcnt,
so a load issued after a store and then waited for with `s_waitcnt vmcnt(0)` also waits for the store's acknowledgement.
A loop `x[i] = x[i] - a[i]` over global memory that the compiler cannot prove alias-free becomes one full round trip per
element (this was ~24 of the 32 us of the covariance downdate's launch).  Prints, per kernel, how many store -> load ->
full-wait sequences its ISA holds.  Usage: store_load_chains.py kernels_*.s (hipcc -S --cuda-device-only output)."""
import re, subprocess, sys

for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"^(_ZN5msckf\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seq = []
        for line in body.splitlines():
            t = line.strip().split(' ')[0] if line.strip() else ''
            if t.startswith(('global_load', 'buffer_load')): seq.append('L')
            elif t.startswith(('global_store', 'buffer_store')): seq.append('S')
            elif t == 's_waitcnt' and 'vmcnt(0)' in line: seq.append('W')
        n = len(re.findall(r"S+L+W", ''.join(seq)))
        if n >= 3:
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
            print(f"{n:4d}  {d}")
