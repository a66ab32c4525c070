import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from msckf_mono_amd import capi, scenario as sc
import helpers as H
for (N,F) in ((12,40),(30,120)):
    nf=N+6
    tr=sc.Trajectory(2,7,N,F,nf)
    res={}
    for name,(route,form) in {"ref":(0,2),"gram":(3,2),"gain":(0,0)}.items():
        bt=capi.Batch(1,N,F,N,capi.F32); bt.set_compression(route); bt.set_covariance_update(form)
        bt.initialize(0,tr.cfg,tr.imu0)
        for k in range(nf): H.device_frame(bt,0,tr,k,N)
        res[name]=(bt.imu_state(0),bt.cam_states(0)[0],bt.covariance(0),bt.last_stats(0)); bt.close()
    for name in ("gram","gain"):
        e=H.state_errors(res[name][0],res["ref"][0],res[name][1],res["ref"][1],res[name][2],res["ref"][2])
        print(N,name,H.worst(e),res[name][3]["n_passed"],res["ref"][3]["n_passed"])
