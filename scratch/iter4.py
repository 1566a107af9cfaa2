import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from active_gs_amd.fused_map_trainer import FusedMapTrainer
from active_gs_amd.facade import SurfelRenderer
from active_gs_amd.synthetic import make_camera, make_room_scene, activate
dev=torch.device('cuda:0')
n,h,w,nf=200_000,512,512,10
raw={k:v.to(dev) for k,v in make_room_scene(n,'office0',seed=0).items()}
frames=[]; a=activate(raw)
for v in range(nf):
    c2w,K=make_camera(v,h,w,focal_px=0.5*512/np.tan(np.pi/6)); c2w,K=c2w.to(dev),K.to(dev)
    with torch.no_grad():
        rr=SurfelRenderer(c2w[None],K[None],(a['means'],raw['harmonics'],a['opacities'],a['confidences'],a['scales'],a['rotations']),torch.zeros(4,device=dev),(0.001,10.),(h,w),dev).render_view_all()
    frames.append(dict(rgb=(rr[0][0]+0.02*torch.randn_like(rr[0][0])).clamp(0,1),depth=rr[1][0].clone(),extrinsic=c2w,intrinsic=K,depth_range=torch.tensor([0.001,10.],device=dev)))
t=FusedMapTrainer({k:v.clone() for k,v in raw.items()},frames,dict(optimization_steps=10,prune_interval=1000))
np.random.seed(0); t.train(steps=30); torch.cuda.synchronize()
