"""ASL (EuRoC mav0) layout: writer/reader round trip, the deterministic runner (asl_msckf.cpp:206-296 call order)
against the scenario-driven loop, ATE; on the GPU the same runner drives the HIP path."""
import numpy as np
import pytest

import helpers as H
from msckf_mono_amd import asl, scenario as sc


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    N, F, nf = 7, 10, 16
    tr = sc.Trajectory(1, 3, N, F, nf)
    root = tmp_path_factory.mktemp("asl")
    mav = asl.write_dataset(str(root), tr)
    return tr, asl.read_dataset(mav)


def test_round_trip(dataset):
    tr, ds = dataset
    assert np.array_equal(ds["readings"][:, :6], tr.readings[:, :6])
    assert np.allclose(ds["readings"][:, 6], tr.dT)                        # Q6: dT = 1/rate_hz
    assert np.allclose(ds["cam"]["q_CI"], tr.cfg["q_CI"], atol=1e-14) and np.allclose(ds["cam"]["p_C_I"], tr.cfg["p_C_I"], atol=1e-15)
    assert len(ds["cam_t"]) == tr.n_frames
    st = tr.stream()
    for k, t in enumerate(ds["cam_t"]):
        e = ds["tracks"].get(int(t), {"cur": ([], []), "new": ([], [])})
        for key in ("cur", "new"):
            assert e[key][1] == [int(i) for i in st[k][key][1]]
            if len(e[key][1]):
                assert np.array_equal(np.array(e[key][0]), np.array(st[k][key][0]))
    s0 = asl.initial_state(ds)
    assert np.allclose(s0[:19], tr.imu0[:19], atol=1e-12)


def test_reader_parses_the_original_euroc_layout():
    """tests/golden/euroc_layout/mav0: a few lines typed in the ORIGINAL EuRoC on-disk layout (not written by asl.write_dataset)
    -- sensor.yaml with the `%YAML:1.0` line cv::FileStorage needs (asl_readers.h:54-63, README), comments, multi-line T_BS
    flow sequences and trailing `#fu, fv, cu, cv`; CSVs with EuRoC's header lines and CRLF line ends -- read the way
    datasets/asl_readers.cpp:12-75, 141-206, 244-342 reads them."""
    import os
    mav = os.path.join(H.ROOT, "tests", "golden", "euroc_layout", "mav0")
    ds = asl.read_dataset(mav)
    # imu0: t, w(3), a(3); dT = 1 / rate_hz whatever the stamps say (asl_readers.cpp:170-171, 202)
    assert ds["imu_t"].tolist() == [1403636579758555392, 1403636579763555584, 1403636579768555520, 1403636579773555456, 1403636579778555392]
    assert ds["readings"].shape == (5, 7) and np.all(ds["readings"][:, 6] == 1.0 / 200)
    assert ds["readings"][0, 0] == -0.099134701513277898 and ds["readings"][4, 5] == -2.484351333333333
    # cam0: q_CI = Quaternion(R_BS).inverse(), p_C_I = p_BS, intrinsics fu fv cu cv (asl_readers.cpp:27-50)
    T = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                  [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                  [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949]])
    assert ds["cam"]["intrinsics"] == [458.654, 457.296, 367.215, 248.375] and ds["cam"]["rate_hz"] == 20
    assert np.allclose(sc.quat_to_rot(ds["cam"]["q_CI"]), T[:, :3].T, atol=1e-9) and np.array_equal(ds["cam"]["p_C_I"], T[:, 3])
    assert ds["cam_t"].tolist() == [1403636579763555584, 1403636579813555456]
    # ground truth: t, p, q(w,x,y,z), v, b_w, b_a; q_IG = q^-1, v <- q * v (asl_readers.cpp:338-339)
    g = ds["gt"]
    assert g["t"][0] == 1403636580838555648 and np.array_equal(g["p"][0], [4.688319, -1.786938, 0.783338])
    q = np.array([0.534108, -0.153029, -0.827383, -0.082152])
    assert np.allclose(g["q_IG"][0], q * [1, -1, -1, -1] / (q @ q), atol=1e-15)
    assert np.allclose(g["v"][0], sc.quat_to_rot(q) @ np.array([-0.027876, 0.033207, 0.800006]), atol=1e-15)
    assert np.array_equal(g["b_g"][2], [-0.003172, 0.021267, 0.078502]) and np.array_equal(g["b_a"][2], [-0.025266, 0.136696, 0.075593])
    assert ds["tracks"] == {}                      # no front-end track dump in an original dataset
    # the runner's parameters come out of the same files (asl_msckf.cpp:73-117): EuRoC's f_u != f_v
    cfg = asl.filter_config_from_dataset(ds)
    assert cfg["u_var_prime"] == (7.0 / 458.654) ** 2 and cfg["v_var_prime"] == (7.0 / 457.296) ** 2


def test_runner_equals_scenario_loop(dataset, oracle_lib):
    po = oracle_lib
    tr, ds = dataset
    st = tr.stream()
    a, b = po.Oracle(po.F64, po.LEAN), po.Oracle(po.F64, po.LEAN)
    out = asl.run(ds, a, tr.cfg)
    b.initialize(tr.cfg, asl.initial_state(ds))
    sid = 0
    for k in range(tr.n_frames):
        b.propagate(ds["readings"][k * sc.IMU_PER_FRAME:(k + 1) * sc.IMU_PER_FRAME]); sid += sc.IMU_PER_FRAME
        b.augmentState(sid, 0.0)
        b.update(st[k]["cur"][0], st[k]["cur"][1]); b.addFeatures(st[k]["new"][0], st[k]["new"][1])
        b.marginalize(); b.pruneEmptyStates()
        assert np.array_equal(out[k][1], b.getImuState())
    e, se, n = asl.ate(out, ds)
    assert n == tr.n_frames and e < 0.02


def test_stage_timing_records_follow_the_reference_message(dataset, oracle_lib, tmp_path):
    """StageTiming (msg/StageTiming.msg): one record per image with the stage names of asl_msckf.cpp:229-296, and the
    timed run gives the same states as the untimed one."""
    po = oracle_lib
    tr, ds = dataset
    cfg = tr.cfg
    a, b = po.Oracle(po.F64, po.LEAN), po.Oracle(po.F64, po.LEAN)
    rec = []
    out_t = asl.run(ds, a, cfg, prune_redundant=True, stage_timing=rec)
    out = asl.run(ds, b, cfg, prune_redundant=True)
    assert len(rec) == len(out) == len(ds["cam_t"])
    for r, (t, _) in zip(rec, out):
        assert r["stamp"] == t and tuple(r["stages"]) == asl.STAGES and all(x >= 0 for x in r["times"])
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(out_t, out))
    p = tmp_path / "stage_timing.csv"
    asl.write_stage_timing(str(p), rec)
    lines = p.read_text().splitlines()
    assert len(lines) == 1 + len(rec) * len(asl.STAGES) and lines[1].split(",")[1] == "imu_prop"


def test_default_parameters_match_the_reference_runner(dataset):
    _, ds = dataset
    cfg = asl.filter_config_from_dataset(ds)
    f_u = ds["cam"]["intrinsics"][0]
    assert cfg["u_var_prime"] == (7.0 / f_u) ** 2 and cfg["max_gn_cost_norm"] == (11.0 / f_u) ** 2   # asl_msckf.cpp:74-78,103-104
    assert cfg["max_cam_states"] == 20 and cfg["min_track_length"] == 3 and cfg["max_track_length"] == 1000   # :116-118
    assert cfg["Q_imu_diag"][0] == 1e-5 and cfg["Q_imu_diag"][3] == 3.6733e-5 and cfg["Q_imu_diag"][6] == 1e-3 and cfg["Q_imu_diag"][9] == 7e-4


def test_standstill_initialisation_of_the_no_ground_truth_runner():
    """asl_msckf_no_ground_truth.cpp:136-173: biases and attitude from the mean IMU reading over a stand-still interval"""
    rng = np.random.default_rng(1)
    # a tilted, stationary IMU: specific force = C_IG (-g), plus biases and noise
    ang = 0.2
    C_IG = np.array([[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]])
    g = np.array([0.0, 0.0, -9.81])
    bg, ba = np.array([0.002, -0.001, 0.003]), np.array([0.02, 0.01, -0.03])
    n = 400
    rd = np.zeros((n, 7)); rd[:, 0:3] = bg + 1e-4 * rng.standard_normal((n, 3)); rd[:, 3:6] = C_IG @ (-g) + ba + 1e-3 * rng.standard_normal((n, 3)); rd[:, 6] = 0.005
    ds = dict(imu_t=np.arange(n, dtype=np.int64) * 5_000_000, readings=rd)
    s = asl.standstill_initial_state(ds, 0, 300 * 5_000_000)
    assert np.allclose(s[4:7], rd[:300, 0:3].mean(0)) and np.all(s[7:10] == 0) and np.all(s[13:16] == 0)   # b_g, v, p
    q = s[0:4]; assert abs(np.linalg.norm(q) - 1) < 1e-12
    R = H.q_to_rot(q)
    am = rd[:300, 3:6].mean(0)
    assert np.allclose(R @ (-g) / 9.81, am / np.linalg.norm(am), atol=1e-12)        # q_IG sends -g onto the measured specific force
    assert np.allclose(s[10:13], R @ g + am) and np.linalg.norm(s[10:13]) < 0.1   # b_a = q_IG g + a_mean: |a_mean| - 9.81 along gravity
    assert np.array_equal(s[19:23], q)                                               # null-space anchors start at the state
    with pytest.raises(ValueError):
        asl.standstill_initial_state(ds, 10**12, 2 * 10**12)


@pytest.mark.gpu
def test_stage_timing_on_the_hip_path(dataset, tmp_path):
    """StageTiming (msg/StageTiming.msg, asl_msckf.cpp:207-212) recorded on the HIP path: one record per image, the
    reference's stage names, a device sync per stage, and the timed run gives bit-identical states to the untimed one."""
    from msckf_mono_amd import capi
    tr, ds = dataset
    rec = []
    a = capi.MSCKF(capi.F32, n_cap=16, f_cap=64, m_cap=16)
    b = capi.MSCKF(capi.F32, n_cap=16, f_cap=64, m_cap=16)
    out_t = asl.run(ds, a, tr.cfg, prune_redundant=True, stage_timing=rec)
    out = asl.run(ds, b, tr.cfg, prune_redundant=True)
    assert len(rec) == len(out) == len(ds["cam_t"])
    for r, (t, _) in zip(rec, out):
        assert r["stamp"] == t and tuple(r["stages"]) == asl.STAGES and all(x > 0 for x in r["times"])
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(out_t, out))
    per_stage = {st: float(np.median([r["times"][i] for r in rec])) for i, st in enumerate(asl.STAGES)}
    assert per_stage["msckf_marginalize"] < 0.05 and per_stage["imu_prop"] < 0.05      # seconds per image, single filter
    asl.write_stage_timing(str(tmp_path / "stage_timing.csv"), rec)
    assert (tmp_path / "stage_timing.csv").read_text().count("\n") == 1 + len(rec) * len(asl.STAGES)


@pytest.mark.gpu
def test_runner_on_the_hip_path(dataset, oracle_lib):
    po = oracle_lib
    from msckf_mono_amd import capi
    tr, ds = dataset
    o = po.Oracle(po.F64, po.LEAN)
    ref = asl.run(ds, o, tr.cfg)
    f = capi.MSCKF(capi.F64, n_cap=16, f_cap=64, m_cap=16)
    got = asl.run(ds, f, tr.cfg)
    for (t1, s1), (t2, s2) in zip(ref, got):
        assert t1 == t2 and H.rel(s2[:16], s1[:16]) < 1e-6
    assert abs(asl.ate(got, ds)[0] - asl.ate(ref, ds)[0]) < 1e-8
