#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 rocpd databases (one --pmc pass per database).
Usage: rocpd_pmc.py out.md db1 [db2 ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x (MI355X_MICROARCH.md, HBM section) -- both the raw and the x2-corrected numbers are printed."""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    rows = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> per-dispatch values (steady-state = upper half)
    for dbp in sys.argv[2:]:
        db = sqlite3.connect(dbp)
        for name, disp, cname, val in db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            name = re.sub(r"\(.*$", "", name); name = re.sub(r"^void ", "", name)
            rows[name][cname].append((disp, val))
    lines = ["| kernel | counter | dispatches | mean per dispatch (steady state) | max |", "|---|---|---|---|---|"]
    res = {}
    for k in sorted(rows):
        for c in sorted(rows[k]):
            by_disp = defaultdict(float)
            for disp, val in rows[k][c]:
                by_disp[disp] += val          # sum over XCDs / shader engines
            vals = [by_disp[d] for d in sorted(by_disp)]
            ss = vals[len(vals) // 2:]        # second half of the run = steady-state window
            mean = sum(ss) / len(ss)
            lines.append("| %s | %s | %d | %.4g | %.4g |" % (k, c, len(vals), mean, max(vals)))
            res[(k, c)] = mean
    txt = "\n".join(lines)
    print(txt)
    open(out, "a").write(txt + "\n")
    # HBM traffic per launch of every kernel that has both counters: mean over the steady-state (second half of the
    # run) dispatches.  Keyed by the bare kernel name (k_feature, k_gram, ...), read by bench.py's roofline block.
    import json, os
    tr = {}
    for k in rows:
        if "FETCH_SIZE" in rows[k] and "WRITE_SIZE" in rows[k]:
            m = re.search(r"(k_[a-zA-Z_0-9]+)", k)
            if not m:
                continue
            f, w = res[(k, "FETCH_SIZE")], res[(k, "WRITE_SIZE")]
            key = m.group(1) + (re.search(r"<(\d+)>", k).group(0) if m.group(1) == "k_gemm_mfma" and re.search(r"<(\d+)>", k) else "")
            if m.group(1) == "k_lit_phase":   # the four phase kernels of the literal anisotropic compression (rows, sweep, basis, elimination)
                ph = re.search(r"k_lit_phase<\w+, *(\d)>", k)
                key += "<%s>" % (ph.group(1) if ph else "?")
            if m.group(1) == "k_chol_mfma":   # two instances per frame: the f64 factorization of Lam^ and the f32 gain solve
                key += "_gram" if "<double" in k else "_gain"
            tr[key] = dict(kernel=k, fetch_kib=f, write_kib=w, bytes_per_launch=(2 * f + w) * 1024,
                           note="FETCH_SIZE doubled (gfx950 under-reports wide coalesced reads by 2x); WRITE_SIZE uncalibrated")
            ex = {c: res[(k, c)] for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                                           "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                                           "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES", "SQ_ACTIVE_INST_LDS",
                                           "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU") if (k, c) in res}
            if ex:   # executed work per launch (wave-level instruction counts summed over the chip), for roofline.executed
                tr[key]["executed"] = ex
    if tr:
        # provenance: which profile round and which commit the passes ran on (bench.py prints it beside the numbers)
        tr["_meta"] = dict(tag=os.environ.get("PMC_TAG", ""), commit=os.environ.get("PMC_COMMIT", ""), csrc_hash=os.environ.get("PMC_CSRC_HASH", ""),
                           command="bench.py --steps 10 --warmup 3 --no-upload-pass --streams 1 under rocprofv3 --pmc (one pass per counter set)")
        json.dump(tr, open(os.path.join(os.path.dirname(out) or ".", "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
