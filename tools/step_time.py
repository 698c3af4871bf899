import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/zero-inflated-gp_amd')
import bench, zigp, torch
X,Y,p=bench.synth(1000000,1024,3)
e=zigp.DenseEngine(0); e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
for prof in (False, True, False):
    e.profile_enable(prof); e.profile_reset()
    e.elbo(p)
    t0=time.time()
    for _ in range(4): e.elbo(p)
    print('profiling %s: %.2f ms/step'%(prof,(time.time()-t0)/4*1e3))
t0=time.time()
for _ in range(4): e.elbo(p, need_grad=False)
print('value only: %.2f ms/step'%((time.time()-t0)/4*1e3))
