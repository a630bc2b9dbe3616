"""debug aid: where does the GPU optimize() deviate from the oracle?"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import nalo_pkg; nalo_pkg.load()
from nalo_slam_amd import synth, binding
import orc
from helpers import rel_err, pose_dist
from test_ba_gpu import make_ctx
win=synth.make_window(w=640,h=480,W=8,P=3000,seed=7)
st6=synth.perturbed_poses(win,sigma_t=0.004,sigma_r=0.0004)
for nb in (False, True):
    res={}
    for kind in ('f32','f64'):
        orc.lib(kind).orc_set_sum_mode(0)
        ba=orc.ba_from_window(win,kind,state6=st6); ba.set_options(6, nb); r=ba.optimize(6)
        res[kind]=([ba.frame(f)['worldToCam'] for f in range(win.W)], ba.slots()[0], ba.frame(win.W-1)['frameEnergyTH'], r, ba.points()['idepth'])
    c=make_ctx(win,st6); r=c.ba_optimize(6, never_break=nb)
    fr,w2c,cal=c.ba_get_frames(); st=c.ba_get_residuals()[0]
    print('never_break',nb,'rmse',r,res['f32'][3],res['f64'][3],'TH',fr[win.W-1].frameEnergyTH,res['f32'][2],res['f64'][2])
    print(' gpu/f32',['%.2e'%pose_dist(w2c[f],res['f32'][0][f]) for f in range(win.W)],'flips',int((st!=res['f32'][1]).sum()))
    print(' f64/f32',['%.2e'%pose_dist(res['f64'][0][f],res['f32'][0][f]) for f in range(win.W)],'flips',int((res['f64'][1]!=res['f32'][1]).sum()))
    print(' gpu/f64',['%.2e'%pose_dist(w2c[f],res['f64'][0][f]) for f in range(win.W)])
    idp=c.ba_get_points()['idepth']; print(' idepth med rel gpu/f32',np.median(np.abs(idp-res['f32'][4])/np.abs(res['f32'][4])),' f64/f32',np.median(np.abs(res['f64'][4]-res['f32'][4])/np.abs(res['f32'][4])))
    c.close()
