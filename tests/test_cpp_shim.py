"""The C++ drop-in shim include/msckf_mono/msckf.h (MSCKF<_S> with the reference's member names, msckf.h:72-848):
compiles without Eigen on CPU; on the GPU box it is linked against libmsckf_hip.so, run with the call order of
datasets/asl_msckf.cpp:227-294 and compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from msckf_mono_amd import scenario as sc

ROOT = H.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp")


def test_shim_compiles_without_eigen(tmp_path):
    out = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-c", SRC, "-o", str(tmp_path / "shim.o")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_shim_eigen_branch_compiles_against_the_reference_types():
    """The branch a maintainer would build (with <Eigen/Dense> and the reference's own types.h): compile-only, against
    the minimal Eigen surface of oracle/ref_shim (Eigen itself is not installed here)."""
    ref_inc = "/root/reference/include"
    if not os.path.exists(os.path.join(ref_inc, "msckf_mono", "types.h")):
        pytest.skip("reference tree not present on this machine")
    src = os.path.join(ROOT, "tests", "cpp", "shim_eigen_check.cpp")
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(ROOT, "include"),
                          "-I" + os.path.join(ROOT, "oracle", "ref_shim"), "-I" + ref_inc, src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


@pytest.mark.gpu
def test_shim_runs_like_the_asl_runner(tmp_path, oracle_lib):
    po = oracle_lib
    from msckf_mono_amd import capi
    exe = str(tmp_path / "shim_demo")
    libdir = os.path.dirname(capi.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + libdir, "-lmsckf_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    N, F, nf = 6, 8, 12
    tr = sc.Trajectory(2, 21, N, F, nf)
    st = tr.stream()
    cam, noise, prm = capi.pack_config(tr.cfg)
    lines = [" ".join(repr(float(x)) for x in np.concatenate([cam, noise, prm, tr.imu0])), str(nf)]
    o = po.Oracle(po.F64, po.LEAN)
    o.initialize(tr.cfg, tr.imu0)
    sid = 0
    for k in range(nf):
        rd = tr.imu_for_frame(k)
        lines.append(str(len(rd)) + " " + " ".join(repr(float(x)) for x in rd.ravel()))
        for kind in ("cur", "new"):
            obs, ids = st[k][kind]
            lines.append(str(len(ids)) + " " + " ".join("%r %r %d" % (float(z[0]), float(z[1]), i) for z, i in zip(obs, ids)))
        o.propagate(rd); sid += len(rd)
        o.augmentState(sid, float(k))
        o.update(st[k]["cur"][0], st[k]["cur"][1]); o.addFeatures(st[k]["new"][0], st[k]["new"][1])
        o.marginalize(); o.pruneEmptyStates()
    run = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr
    rows = run.stdout.strip().splitlines()
    imu = np.array([float(x) for x in rows[0].split()])
    ncam, nmap = [int(x) for x in rows[1].split()]
    assert ncam == o.getNumCamStates() and nmap == len(o.getMap())
    assert H.rel(imu, o.getImuState()[:16]) < 1e-6
    assert abs(float(rows[2]) - np.trace(o.getCovariance())) / np.trace(o.getCovariance()) < 1e-6
    # getCamStates(): ids, times, tracked_feature_ids.size(), last_correlated_id  (asl_msckf.cpp:384-388)
    cs = rows[3].split()
    assert int(cs[0]) == ncam
    tm, nt, lc = o.getCamMeta()
    ids = o.getCamStates()[1]
    for i in range(ncam):
        sid, t, k, l = int(cs[1 + 4 * i]), float(cs[2 + 4 * i]), int(cs[3 + 4 * i]), int(cs[4 + 4 * i])
        assert (sid, k, l) == (int(ids[i]), int(nt[i]), int(lc[i])) and t == tm[i]
    # getPrunedStates(): pose + time of every pruned state, sorted by id  (asl_msckf.cpp:409-424)
    ps = rows[4].split()
    ref = o.getPrunedStates()
    assert int(ps[0]) == len(ref) > 0
    got = np.array([float(x) for x in ps[1:]]).reshape(-1, 9)
    assert np.array_equal(got[:, 0], ref[:, 8]) and np.array_equal(got[:, 1], ref[:, 7])
    assert np.allclose(got[:, 2:9], ref[:, 0:7], atol=1e-8)
