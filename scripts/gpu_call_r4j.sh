cd /root/repo
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_literal.py tests/test_cpp_shim.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
tail -4 $O/pytest.txt
MSCKF_HIP_LITERAL_TIMERS=1 python - <<PY 2>&1 | tail -14
import sys, numpy as np, time
sys.path.insert(0, "/root/repo")
from msckf_mono_amd import capi, scenario as sc
import bench
for B in (8, 128):
    c = dict(bench.CONFIGS["cfg4"]); c["B"] = B
    nfr = 36
    trajs = bench.make_trajectories(c, 0, nfr)
    bt = capi.Batch(B, 30, 200, 30, capi.F32)
    bt.scenario_alloc(nfr, 10)
    for b, tr in enumerate(trajs):
        bt.initialize(b, tr.cfg, tr.imu0)
        for f in range(nfr):
            fr = tr.frames[f]
            bt.scenario_set(f, b, tr.imu_for_frame(f), fr["M"], fr["slots"], fr["obs"], 1 if fr["Nw"] == 30 else 0)
    bt.scenario_commit()
    bt.run_frames(0, 32); bt.sync()
    for f in range(32, 36):
        t0 = time.perf_counter(); bt.run_frames(f, f + 1); bt.sync(); dt = time.perf_counter() - t0
        infos = [bt.L.msckf_hip_literal_info and None for _ in range(0)]
        import ctypes as C
        routes = []
        for b in range(B):
            o = np.zeros(8, dtype=np.int32); bt.L.msckf_hip_literal_info(bt.h, b, o.ctypes.data_as(C.POINTER(C.c_int))) if b else None
            routes.append(int(o[4]))
        print("B", B, "frame", f, "ms %.2f" % (dt * 1e3), "routes (b>0): fast", routes.count(1), "general", routes.count(2))
    print(bt.literal_info(0))
    bt.close()
PY
timeout 600 python bench.py --config cfg2 --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<PY
import json
j = json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", round(j["value"]), j["latency_us"]["stage_mean"], j["repeats"]["values"])
PY
