#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/ from DATA files of the
reference checkout (run once in the build container; /root/reference does not
exist on the GPU box, so tests only ever read the .npz files written here).

Only numeric data is extracted (inputs and stored results held by the reference's
MAT v5 files); no reference source text is read or stored.

  arm_data.npz        datafiles/arm-3link-markers-noload-50trials_train-10_val-5.mat
                      train{1..10}.{t,y,u} merged + trial lengths, val{1}.{t,y,u}
  arm_blockM.npz      systems/.../simulations/blockM_.../{bilinear,linear,nonlinear}*.mat
                      res_bilin / res_lin: Y, U, R, Z (300x34), comp_time;
                      res_nonlin: Z width (88)  -> pins scale+pairs+monomial order+pca+econ lift
  blockM_ref.npz      trajectories/files/blockM_c0p45-0p35_0p5x0p5_15sec.mat  ref.y (301x2), Ts
  rand_systems.npz    datafiles/rand-systems_2021-01-10_16-59 (1)/rsys-all_*.mat, first 3 systems
  arm_circle.npz      systems/.../simulations/circle_c0-0p7_r0p3_15sec/bilinear_..._2020-06-09_16-43.mat  res{1..3}: three more closed
                      loops of the same N = 34 bilinear model under loads (Y, U, R, Z 300x34, W) - stored MATLAB inputs, slope
                      constant 1e-2; ..._2020-06-21_23-31.mat  res_loaded{1..3}: Z (300x96 = [1; w] (x) psi, N = 32), What, W
  arm_plant.npz       the plant behind the stored closed loops: train{1}.params (numeric fields),
                      res_bilin.{X,U,Y,err} (X(k+1) = Arm.simulate_Ts(X(k), U(k)), Ksim.m:239-245),
                      train{1}.{x,u,y} rows 1..200
"""
import glob
import os

import numpy as np
import scipy.io as sio

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path):
    return sio.loadmat(os.path.join(REF, path), squeeze_me=False, struct_as_record=False)


def main():
    d = load('datafiles/arm-3link-markers-noload-50trials_train-10_val-5.mat')
    tr = [d['train'][0, i][0, 0] for i in range(d['train'].shape[1])]
    va = d['val'][0, 0][0, 0]
    np.savez_compressed(
        os.path.join(OUT, 'arm_data.npz'),
        train_t=np.vstack([t.t for t in tr]), train_y=np.vstack([t.y for t in tr]),
        train_u=np.vstack([t.u for t in tr]), train_len=np.array([t.t.shape[0] for t in tr]),
        val_t=va.t, val_y=va.y, val_u=va.u)

    base = ('systems/thesis-arm-markers_noload_3-mods_1-links_20hz/simulations/'
            'blockM_c0p45-0p35_0p5x0p5_15sec/')
    rb = load(base + 'bilinear_poly-3_n-6_m-3_del-0_2020-06-09_16-43.mat')['res_bilin'][0, 0]
    rl = load(base + 'linear_poly-3_n-6_m-3_del-0_2020-06-09_16-42.mat')['res_lin'][0, 0]
    rn = load(base + 'nonlinear_poly-3_n-6_m-3_del-0_2020-06-13_14-10.mat')['res_nonlin'][0, 0]
    np.savez_compressed(
        os.path.join(OUT, 'arm_blockM.npz'),
        bilin_Y=rb.Y, bilin_U=rb.U, bilin_R=rb.R, bilin_Z=rb.Z, bilin_comp_time=rb.comp_time,
        lin_Y=rl.Y, lin_U=rl.U, lin_Z=rl.Z, lin_comp_time=rl.comp_time,
        nonlin_Zwidth=np.array(rn.Z.shape[1]), nonlin_comp_time=rn.comp_time)

    cbase = ('systems/thesis-arm-markers_noload_3-mods_1-links_20hz/simulations/circle_c0-0p7_r0p3_15sec/')
    rc = load(cbase + 'bilinear_poly-3_n-6_m-3_del-0_2020-06-09_16-43.mat')['res']
    rl3 = load(cbase + 'bilinear_poly-3_n-6_m-3_del-0_2020-06-21_23-31.mat')['res_loaded']
    circ = {}
    for i in range(3):
        a = rc[0, i][0, 0]
        for f in ('Y', 'U', 'R', 'Z', 'W', 'err'):
            circ[f'run{i}_{f}'] = getattr(a, f)
        b = rl3[0, i][0, 0]
        for f in ('Z', 'What', 'W', 'Y', 'U'):
            circ[f'loaded{i}_{f}'] = getattr(b, f)
    np.savez_compressed(os.path.join(OUT, 'arm_circle.npz'), **circ)

    r = load('trajectories/files/blockM_c0p45-0p35_0p5x0p5_15sec.mat')['ref'][0, 0]
    np.savez_compressed(os.path.join(OUT, 'blockM_ref.npz'), y=r.y, Ts=r.Ts, t=r.t)

    f = sorted(glob.glob(os.path.join(REF, 'datafiles/rand-systems_2021-01-10_16-59 (1)/rsys-all_*.mat')))[0]
    a = sio.loadmat(f, squeeze_me=False, struct_as_record=False)['data4sysid_all']
    out = {}
    for i in range(3):
        s = a[i, 0][0, 0]
        trs = [s.train[0, j][0, 0] for j in range(s.train.shape[1])]
        v = s.val[0, 0][0, 0]
        out[f's{i}_train_t'] = np.vstack([t.t for t in trs])
        out[f's{i}_train_y'] = np.vstack([t.y for t in trs])
        out[f's{i}_train_u'] = np.vstack([t.u for t in trs])
        out[f's{i}_val_t'], out[f's{i}_val_y'], out[f's{i}_val_u'] = v.t, v.y, v.u
    np.savez_compressed(os.path.join(OUT, 'rand_systems.npz'), **out)
    p = tr[0].params[0, 0]
    par = {'p_' + f: np.asarray(getattr(p, f), dtype=np.float64).squeeze() for f in p._fieldnames if f != 'sysName'}
    np.savez_compressed(os.path.join(OUT, 'arm_plant.npz'), bilin_X=rb.X, bilin_U=rb.U, bilin_Y=rb.Y, bilin_err=rb.err,
                        lin_err=rl.err, nonlin_err=rn.err,
                        train_x=tr[0].x[:200], train_u=tr[0].u[:200], train_y=tr[0].y[:200], **par)
    for fn in sorted(glob.glob(os.path.join(OUT, '*.npz'))):
        print(fn, os.path.getsize(fn))


if __name__ == '__main__':
    main()
