#!/usr/bin/env python3
"""Merge the PMC summary of the cfg4 (literal anisotropic route) passes into the main pmc_traffic.json (cfg3 passes):
   pmc_merge.py main/pmc_traffic.json lit/pmc_traffic.json
adds the literal route's kernels (k_lit_pre, k_lit_gamma, k_lit_phase<0..3>) and `k_literal` = the sum over the four phase
kernels (what bench.py's stage timer `literal` brackets).  Other kernels of the cfg4 passes are left out (the cfg3 passes
have them at the headline configuration)."""
import json
import sys

main_p, lit_p = sys.argv[1], sys.argv[2]
main = json.load(open(main_p)); lit = json.load(open(lit_p))
tot = None
for k, v in lit.items():
    if not k.startswith("k_lit"):
        continue
    main[k] = dict(v, config="cfg4 (128 trajectories, literal anisotropic route)")
    if k.startswith("k_lit_phase"):
        if tot is None:
            tot = dict(kernel="k_lit_phase<., 0..3> (sum of the four launches)", fetch_kib=0.0, write_kib=0.0, bytes_per_launch=0.0, note=v.get("note", ""), executed={},
                       config="cfg4 (128 trajectories, literal anisotropic route)")
        for f in ("fetch_kib", "write_kib", "bytes_per_launch"):
            tot[f] += v[f]
        for c, x in (v.get("executed") or {}).items():
            tot["executed"][c] = tot["executed"].get(c, 0.0) + x
if tot is not None:
    main["k_literal"] = tot
json.dump(main, open(main_p, "w"), indent=1)
print("merged:", sorted(k for k in main if k.startswith("k_lit")))
