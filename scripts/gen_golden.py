#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the INDEPENDENT numpy/scipy restatement (oracle/np_oracle.py).

The reference ships no golden vectors and cannot be built here, so these fixtures pin
the C++ oracle and the HIP path against a second implementation instead.  Inputs come from the seeded
generator (msckf_mono_amd/scenario.py); everything needed to replay them is stored in the file.
Run:  python scripts/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import np_oracle as npo  # noqa: E402
from msckf_mono_amd import scenario as sc  # noqa: E402


def worklist_case(name, config_id, traj, N, F, nf, **kw):
    tr = sc.Trajectory(config_id, traj, N, F, nf, **kw)
    f = npo.NpMSCKF(tr.cfg, tr.imu0, nullspace="svd")
    imu, P, cams, ncam, dx = [], [], [], [], []
    Dmax = 15 + 6 * N
    for k in range(nf):
        for rd in tr.imu_for_frame(k):
            f.propagate(rd)
        f.augment(k)
        fr = tr.frames[k]
        if len(fr["M"]):
            f.set_tracks(fr["M"], fr["slots"], fr["obs"])
            f.marginalize()
        if len(f.cams) == N:
            f.drop_oldest(1)
        imu.append(f.imu29())
        Pp = np.zeros((Dmax, Dmax)); Pp[:f.P.shape[0], :f.P.shape[0]] = f.P; P.append(Pp)
        cc = np.zeros((N, 7)); cc[:len(f.cams)] = f.cam_array(); cams.append(cc)
        ncam.append(len(f.cams))
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, config_id=config_id, traj=traj, N=N, F=F, nf=nf, imu=np.array(imu), P=np.array(P),
                        cams=np.array(cams), ncam=np.array(ncam))
    print("wrote", out, os.path.getsize(out), "bytes")


def stream_case(name, config_id, traj, N, F, nf):
    """Bookkeeping path: MSCKF::update / addFeatures / marginalize / pruneEmptyStates (asl_msckf.cpp:269-294)."""
    tr = sc.Trajectory(config_id, traj, N, F, nf)
    st = tr.stream()
    f = npo.NpMSCKF(tr.cfg, tr.imu0, nullspace="svd")
    imu, ncam, nres = [], [], []
    for k in range(nf):
        for rd in tr.imu_for_frame(k):
            f.propagate(rd)
        f.augment(k)
        f.update(st[k]["cur"][0], st[k]["cur"][1])
        f.add_features(st[k]["new"][0], st[k]["new"][1])
        f.marginalize()
        f.prune_empty()
        imu.append(f.imu29()); ncam.append(len(f.cams)); nres.append(len(f.to_resid))
    D = f.P.shape[0]
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(out, config_id=config_id, traj=traj, N=N, F=F, nf=nf, imu=np.array(imu), ncam=np.array(ncam),
                        nres=np.array(nres), P_final=f.P, cams_final=f.cam_array())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    worklist_case("worklist_n6_f10", 2, 11, 6, 10, 12)
    worklist_case("worklist_n10_f50", 2, 0, 10, 50, 14)
    stream_case("stream_n6_f8", 2, 21, 6, 8, 14)
