import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import helpers as H, pyoracle as po
from msckf_mono_amd import capi, scenario as sc
import test_gpu_literal as T
N, F, nf, n_upd = 30, 200, 42, 10
tr = T._aniso(N, F, nf, 0, cfgid=3)
t = po.Oracle(po.F32, po.GRAM); t.setWhiten(True); t.initialize(tr.cfg, tr.imu0)
first = nf - n_upd
for k in range(first): H.oracle_frame(t, tr, k, N)
o = po.Oracle(po.F32, po.LEAN); o.setTinyRowTol(8e-4); o.initialize(tr.cfg, tr.imu0)
while o.getNumCamStates() < t.getNumCamStates(): o.augmentState(o.getNumCamStates(), 0.0)
T._force(o, t)
bt = capi.Batch(1, N, F, 32, capi.F32); bt.initialize(0, tr.cfg, tr.imu0)
for _ in range(o.getNumCamStates()): bt.augment_range(0, 1)
for k in range(first, nf):
    H.copy_oracle_to_device(o, bt, 0)
    H.oracle_frame(o, tr, k, N); H.device_frame(bt, 0, tr, k, N)
    info = bt.literal_info(0); so = o.lastStats(); sd = bt.last_stats(0)
    e = T._errs(bt, 0, o)
    print(k, "mrej", so["n_motion_rejected"], "kept", info["kept_rows"], so["r_rows"], "passed", sd["n_passed"], so["n_passed"], "handed", info.get("handed"), {a: float("%.2e" % b) for a, b in e.items()})
