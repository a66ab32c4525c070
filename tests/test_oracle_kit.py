"""Known-answer and invariant tests of the oracle's linear-algebra kit and of the filter invariants
(SURVEY.md section 4): chi2 table, covariance symmetry after every call, float-vs-double agreement."""
import numpy as np
import pytest
from scipy.stats import chi2

import helpers as H
from msckf_mono_amd import scenario as sc


def test_chi2_table_pinned():
    """msckf.h:91-95: table[i-1] = quantile(chi2(i), 0.05); values pinned in SURVEY.md section 8c."""
    import re
    txt = open(H.ROOT + "/oracle/chi2_table.h").read()
    vals = [float(x) for x in re.findall(r"^\s+([0-9.eE+-]+),?$", txt, flags=re.M)]
    assert len(vals) == 99
    for k, v in {1: 0.003932140000019522, 2: 0.10258658877510106, 3: 0.35184631774927144, 31: 19.280568559129293,
                 32: 20.071913464548288, 99: 77.04633186376029}.items():
        assert abs(vals[k - 1] - v) < 1e-13 * max(1, v)
    assert np.allclose(vals, chi2.ppf(0.05, np.arange(1, 100)), rtol=1e-13)
    dev = open(H.ROOT + "/msckf_mono_amd/csrc/chi2_table.h").read()
    assert [float(x) for x in re.findall(r"^\s+([0-9.eE+-]+),?$", dev, flags=re.M)] == vals


def test_covariance_symmetric_and_psd(oracle_lib):
    po = oracle_lib
    tr = sc.Trajectory(2, 4, 8, 25, 18)
    o = po.Oracle(po.F64, po.LEAN)
    o.initialize(tr.cfg, tr.imu0)
    for k in range(18):
        o.propagate(tr.imu_for_frame(k))
        P = o.getCovariance(); assert np.array_equal(P, P.T)
        o.augmentState(k, 0)
        P = o.getCovariance(); assert np.array_equal(P, P.T)
        fr = tr.frames[k]
        if len(fr["M"]):
            o.setTracks(fr["M"], fr["slots"], fr["obs"]); o.marginalize()
            P = o.getCovariance(); assert np.array_equal(P, P.T)
            assert np.linalg.eigvalsh(P).min() > -1e-12
        if o.getNumCamStates() == 8:
            o.dropOldest(1)
            P = o.getCovariance(); assert np.array_equal(P, P.T)


def test_float_oracle_tracks_double(oracle_lib):
    """teacher-forced float vs double oracle: within the 1e-3 float bar of BASELINE.json"""
    po = oracle_lib
    N, F, nf = 10, 40, 16
    tr = sc.Trajectory(2, 9, N, F, nf)
    d, f = po.Oracle(po.F64, po.LEAN), po.Oracle(po.F32, po.LEAN)
    d.initialize(tr.cfg, tr.imu0); f.initialize(tr.cfg, tr.imu0)
    for k in range(nf):
        if k:
            f.setCovariance(d.getCovariance()); f.setImuState(d.getImuState())
            for i, c in enumerate(d.getCamStates()[0]):
                f.setCamPose(i, c)
            f.setNumResidualized(d.numResidualized())
        H.oracle_frame(d, tr, k, N); H.oracle_frame(f, tr, k, N)
        e = H.state_errors(f.getImuState(), d.getImuState(), f.getCamStates()[0], d.getCamStates()[0], f.getCovariance(), d.getCovariance())
        assert H.worst(e) < 1e-3, (k, e)


def test_gate_uses_5pct_quantile_of_chi2_M_plus_1(oracle_lib):
    """Q3: with true pixel noise == assumed noise virtually every track is rejected; with the scenario's
    0.5 px vs 7 px the pass rate is > 95 %."""
    po = oracle_lib
    N, F, nf = 8, 40, 12
    for obs_px, lo, hi in ((0.5, 0.95, 1.01), (7.0, -0.01, 0.2)):
        tr = sc.Trajectory(2, 2, N, F, nf, obs_noise_px=obs_px)
        o = po.Oracle(po.F64, po.LEAN)
        o.initialize(tr.cfg, tr.imu0)
        for k in range(nf):
            H.oracle_frame(o, tr, k, N)
        s = o.lastStats()
        tri_ok = s["n_tracks"] - s["n_motion_rejected"] - s["n_tri_rejected"]
        rate = s["n_passed"] / max(tri_ok, 1)
        assert lo < rate < hi, (obs_px, s)


def test_first_four_tracks_skip_check_motion(oracle_lib):
    """Q4: num_feature_tracks_residualized_ > 3 gates checkMotion (msckf.h:354)."""
    po = oracle_lib
    N, F = 6, 6
    cfg = sc.filter_config(N)
    cfg["translation_threshold"] = 1e3        # every track fails checkMotion
    tr = sc.Trajectory(2, 5, N, F, 6, cfg=cfg)
    o = po.Oracle(po.F64, po.LEAN)
    o.initialize(tr.cfg, tr.imu0)
    for k in range(4):
        H.oracle_frame(o, tr, k, N)
    s = o.lastStats()
    assert s["n_tracks"] == F and s["n_passed"] + s["n_gate_rejected"] + s["n_tri_rejected"] == 4 and s["n_motion_rejected"] == F - 4
    assert o.numResidualized() == 4 - s["n_tri_rejected"]


def test_gram_mode_follows_the_qr_modes_over_a_long_float_run(oracle_lib):
    """GRAM mode = the HIP library's compression route on the CPU: [T_H | r_n] = chol([H_o|r_o]^T [H_o|r_o]) accumulated in
    double with pivot skipping.  Free-running for 200 frames it stays with the Householder modes -- to rounding in double,
    to float noise in float -- and reports the unobservable directions as missing rows."""
    import numpy as np
    import helpers as H
    from msckf_mono_amd import scenario as sc
    po = oracle_lib
    N, F, nf = 8, 30, 200
    tr = sc.Trajectory(2, 77, N, F, nf)
    E = lambda x, y: H.state_errors(x.getImuState(), y.getImuState(), x.getCamStates()[0], y.getCamStates()[0], x.getCovariance(), y.getCovariance())
    fs = dict(d=po.Oracle(po.F64, po.LEAN), gd=po.Oracle(po.F64, po.GRAM), a=po.Oracle(po.F32, po.LEAN), g=po.Oracle(po.F32, po.GRAM))
    for f in fs.values():
        f.initialize(tr.cfg, tr.imu0)
    for k in range(nf):
        for f in fs.values():
            H.oracle_frame(f, tr, k, N)
    assert H.worst(E(fs["gd"], fs["d"])) < 1e-7                        # double: to rounding
    # float: 200 free-running frames wander by float noise (~1e-2 on the accelerometer bias for EITHER route); the
    # information form must not wander further from the double filter than the Householder route does
    e_qr, e_gram = E(fs["a"], fs["d"]), E(fs["g"], fs["d"])
    assert H.worst(e_qr) < 3e-2 and H.worst(e_gram) < 3e-2, (e_qr, e_gram)
    assert H.worst(e_gram) < 3 * H.worst(e_qr) + 1e-3, (e_qr, e_gram)
    for a, g in ((fs["d"], fs["gd"]), (fs["a"], fs["g"])):
        assert 0 < g.lastStats()["r_rows"] < a.lastStats()["r_rows"]
        assert g.lastStats()["m_rows"] == a.lastStats()["m_rows"]


def test_restatement_is_clean_under_asan_and_ubsan(tmp_path):
    """The restatement (oracle/msckf_oracle.hpp) built with -fsanitize=address,undefined (make -C oracle asan) runs a filter
    scenario -- work-list path with pruning, the public id-stream path with pruneRedundantStates, float and double -- in a
    child process with the sanitizer runtimes preloaded; any out-of-bounds access, use-after-free or undefined operation
    aborts the child (-fno-sanitize-recover)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle")); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, helpers as H, pyoracle as po
from msckf_mono_amd import scenario as sc
assert po.lib()._name.endswith("liboracle_asan.so")
for dt in (po.F64, po.F32):
    N, F, nf = 8, 24, 16
    tr = sc.Trajectory(2, 7, N, F, nf)
    for mode in (po.LEAN, po.FAITHFUL):
        o = po.Oracle(dt, mode); o.initialize(tr.cfg, tr.imu0)
        for k in range(nf):
            H.oracle_frame(o, tr, k, N)
        assert np.all(np.isfinite(o.getCovariance()))
N, F, nf = 26, 12, 30
cfg = sc.filter_config(N); cfg["max_cam_states"] = 20; cfg["redundancy_distance_thresh"] = 0.25; cfg["redundancy_angle_thresh"] = 0.25; cfg["translation_threshold"] = 0.01
tr = sc.Trajectory(2, 77, N, F, nf, cfg=cfg); st = tr.stream()
o = po.Oracle(po.F64, po.LEAN); o.initialize(tr.cfg, tr.imu0)
for k in range(nf):
    o.propagate(tr.imu_for_frame(k)); o.augmentState(k, tr.frame_times[k]); o.update(*st[k]["cur"]); o.addFeatures(*st[k]["new"])
    o.marginalize(); o.pruneRedundantStates(); o.pruneEmptyStates()
o.finish()
print("SANITIZED_OK")
''' % (root, root, root)
    env = dict(os.environ, ORACLE_SANITIZED="1", LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZED_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
