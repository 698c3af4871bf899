"""Diagnostic: where the BK steps of the GEMM core spend their cycles (wave 0 of every workgroup, s_memtime stamps).
Needs a stamp build:  git apply tools/stamp_build.patch && python zero-inflated-gp_amd/zigp/build.py &&
cp zero-inflated-gp_amd/lib/libzigp.so zero-inflated-gp_amd/lib/libzigp_stamp.so && git checkout zero-inflated-gp_amd/csrc
(stamps never go into the production library).  Result of round 1: profiles/r01d_gemm_stamps.txt"""
import sys, os, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/zero-inflated-gp_amd')
os.environ['ZIGP_LIB']='/root/repo/zero-inflated-gp_amd/lib/libzigp_stamp.so'
import bench, zigp, torch
X,Y,p=bench.synth(262144,1024,3)
e=zigp.DenseEngine(0); e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
e.elbo(p)
e.lib.zigp_debug_stamps.argtypes=[C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
e.lib.zigp_debug_stamps(e.ctx, 1, None)
e.elbo(p)
out=(C.c_ulonglong*48)()
e.lib.zigp_debug_stamps(e.ctx, 0, out)
for i,n in enumerate(['A1','A2','H','J','syrk','other']):
    vm,bar,iss,comp,loop,allc,iters,tiles=[out[8*i+j] for j in range(8)]
    if tiles==0: continue
    print('%-5s tiles %6d iters/tile %5.1f  cycles/iter %7.0f : vmcnt %5.1f%% barrier %5.1f%% issue %4.1f%% compute %5.1f%% | loop/total %.3f'%(n,tiles,iters/tiles,loop/iters,100*vm/loop,100*bar/loop,100*iss/loop,100*comp/loop,loop/allc))
