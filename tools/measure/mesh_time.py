import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sofima_amd import mesh
for shape in [(2,1,32,32),(2,1,32,64),(2,1,64,64),(2,1,128,128),(2,1,205,205),(2,4,205,205)]:
    rng=np.random.default_rng(0)
    prev=(rng.standard_normal(shape)*5).astype(np.float32)
    cfg=mesh.IntegrationConfig(dt=0.001,gamma=0.0,k0=0.01,k=0.1,stride=(40,40),num_iters=1000,max_iters=1000,stop_v_max=1e-9,dt_max=1000,start_cap=0.01,final_cap=10,prefer_orig_order=True)
    x=torch.zeros(shape,device='cuda'); pv=torch.from_numpy(prev).cuda()
    mesh.relax_mesh(x,pv,cfg); torch.cuda.synchronize()
    t=time.perf_counter(); mesh.relax_mesh(x,pv,cfg); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(shape, 'us/step %.2f'%(dt*1e3), 'Gupd/s %.2f'%(np.prod(shape[1:])*1000/dt/1e9))
