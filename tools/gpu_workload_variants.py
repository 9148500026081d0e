import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
n=4096; t=model.load_asset("humanoid3d_walk")
def run(label, actions=None, open_loop=False, pack=0):
    env=BatchEnv(t,n,seed=1,wave_packing=pack); stream=torch.cuda.current_stream().cuda_stream; env.set_stream(stream); env.reset()
    dev=torch.device("cuda")
    st=torch.zeros((n,env.S),dtype=torch.float32,device=dev); rw=torch.zeros(n,dtype=torch.float32,device=dev)
    tm=torch.zeros(n,dtype=torch.int32,device=dev); vd=torch.zeros(n,dtype=torch.int32,device=dev); en=torch.zeros(n,dtype=torch.int32,device=dev)
    ac=None if actions is None else torch.from_numpy(actions).to(dev)
    ends=0
    for ph in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for k in range(100):
            env.step_device(0 if ac is None else ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True, open_loop=open_loop)
            if ph==1: ends+=int(en.sum().item())
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(label, "ms/step %.3f"%(1e3*dt/100), "episode ends per step %.1f"%(ends/100), "mean reward %.3f"%rw.mean().item(), flush=True)
offs=None
run("open-loop A1 duo", None, True)
run("zeros A0 duo", np.zeros((n,28),np.float32))
run("zeros A0 single", np.zeros((n,28),np.float32), pack=1)
env=BatchEnv(t,1); o=env.offsets_scales(); amean=(-o["action_offset"]).astype(np.float32)
run("action = a_mean duo", np.tile(amean,(n,1)))
