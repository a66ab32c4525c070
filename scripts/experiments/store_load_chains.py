#!/usr/bin/env python3
"""Static audit of the device code for serialized read-modify-write chains: on gfx9-family parts stores count in vmcnt,
so a load issued after a store and then waited for with `s_waitcnt vmcnt(0)` also waits for the store's acknowledgement.
A loop `x[i] = x[i] - a[i]` over global memory that the compiler cannot prove alias-free becomes one full round trip per
element (the covariance downdate's epilogue, k_gram's block-diagonal term and the state injection were written that way).
Prints, per kernel, how many store -> load -> full-wait sequences its ISA holds.
Usage: store_load_chains.py kernels_*.s   (hipcc -S --cuda-device-only output; tests/test_information_form_math.py keeps the
main path's epilogues free of such chains)."""
import re
import subprocess
import sys


def count_chains(path):
    """{mangled kernel name: number of store -> load -> s_waitcnt vmcnt(0) sequences}"""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_ZN5msckf\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seq = []
        for line in body.splitlines():
            t = line.strip().split(' ')[0] if line.strip() else ''
            if t.startswith(('global_load', 'buffer_load')):
                seq.append('L')
            elif t.startswith(('global_store', 'buffer_store')):
                seq.append('S')
            elif t == 's_waitcnt' and 'vmcnt(0)' in line:
                seq.append('W')
        out[name] = len(re.findall(r"S+L+W", ''.join(seq)))
    return out


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for name, n in count_chains(path).items():
            if n >= 3:
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
                print(f"{n:4d}  {d}")
