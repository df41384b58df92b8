import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
import bench
ctx=kra.Context(0)
a,b,u=bench.synth_pairs(100000)
basis=kra.Basis(ctx,"bilinear",6,3,[("poly",kra.poly_exponent_table(6,3)[6:])])
snaps=kra.Snapshots(ctx,a,b,u)
mpc,setup=bench.mpc_problem(kra,ctx,basis,snaps)
zeta,up,Yr=bench.mpc_inputs(50)
for i in range(8):
    U,z,st=mpc.step_zeta(basis,zeta[i],up[i],Yr[i])
    us=np.zeros(8); cnt=(C.c_int*2)()
    F.lib().kp_mpc_last_profile(mpc.handle,F.dptr(us),cnt)
    print(st, np.round(us[:6],1), cnt[0], cnt[1], 'kernel_us', round(ctx.timer(2)*1e3,1))
