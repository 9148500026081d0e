import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, parity_common as pc
lib='/root/repo/deepmimic_amd/csrc/libdm_hip.so'
t0s=[0.0,0.37,0.8,0.11,0.5,0.9]
for prec in (64,32):
    dr,ds,ok=pc.batch_rollout_compare("humanoid3d_walk",prec,lib,steps=10,t0s=t0s,wave_packing=2)
    print("duo",prec,"dr",["%.1e"%x for x in dr],ok)
dr,ds,ok=pc.batch_rollout_compare("humanoid3d_walk",64,lib,steps=2,t0s=[0.0,0.4,0.2,0.6],wave_packing=2,lifts=[-0.3,0.0,0.0,-0.25]); print("heavy",dr.max(),ds.max(),ok)
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
for wp in (1,2):
    env=BatchEnv(model.load_asset("humanoid3d_walk"),4096,seed=1234,test_mode=True,wave_packing=wp); env.reset(); env.bench_rollout(20,1)
    ms=env.bench_rollout(0,100)/100; print("wave_packing",wp,"ms/step",ms,"env-steps/s",4096/ms*1e3)
