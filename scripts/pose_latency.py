import sys, time, json
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rpg_svo_amd import synth, tracking, capi
dev=torch.device('cuda:0')
lib=capi.load()
cam=synth.Camera(752,480,315.5,315.5,376.0,240.0)
rng=np.random.default_rng(0); torch.manual_seed(0)
N=120
T=synth.make_trajectory(2, seed=1)
px=np.stack([rng.uniform(60,690,N), rng.uniform(60,420,N)],-1)[None]
f,pos=synth.features_3d(T[:1], cam, torch.as_tensor(px))
t=lambda a,dt: torch.as_tensor(np.ascontiguousarray(a),dtype=dt,device=dev)
from rpg_svo_amd import se3
Tn=se3.mul(se3.exp(rng.normal(size=(1,6))*2e-3), T[:1])
args=(cam, t([N],torch.int32), f.to(dev), torch.zeros(1,N,dtype=torch.int32,device=dev), (pos+0.002*torch.randn(pos.shape,dtype=torch.float64)).to(dev), torch.ones(1,N,dtype=torch.uint8,device=dev), t(Tn,torch.float64))
import ctypes as C
cam_c = capi.camera(cam)
n_t, f_t, lvl_t, pos_t, has_t, T_t = args[1:]
T_work = T_t.clone(); has_work = has_t.clone()
Cov = torch.zeros(1,36,dtype=torch.float64,device=dev); stats = torch.zeros(1,4,dtype=torch.float64,device=dev); ran = torch.zeros(1,dtype=torch.int32,device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def call():
    T_work.copy_(T_t); has_work.copy_(has_t)
    capi.check(lib.svo_hip_pose_optimize(C.byref(cam_c), 1, n_t.data_ptr(), N, f_t.data_ptr(), lvl_t.data_ptr(), pos_t.data_ptr(), has_work.data_ptr(), 2.0, 10, T_work.data_ptr(), Cov.data_ptr(), stats.data_ptr(), ran.data_ptr(), st))
for _ in range(5): call()
torch.cuda.synchronize()
for n_iter in (0, 1, 2, 5, 10):
    ts=[]
    for _ in range(60):
        T_work.copy_(T_t); has_work.copy_(has_t)
        s_=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s_.record()
        capi.check(lib.svo_hip_pose_optimize(C.byref(cam_c), 1, n_t.data_ptr(), N, f_t.data_ptr(), lvl_t.data_ptr(), pos_t.data_ptr(), has_work.data_ptr(), 2.0, n_iter, T_work.data_ptr(), Cov.data_ptr(), stats.data_ptr(), ran.data_ptr(), st))
        e.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e)*1e3)
    print("pose_opt_kernel, one frame of %d observations, n_iter=%d: median %.1f us, min %.1f us"%(N, n_iter, np.median(ts), np.min(ts)))
