import sys, torch, numpy as np
sys.path.insert(0, ".")
import os
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.synthetic import make_camera, make_room_scene, activate
from active_gs_amd.camera import camera_matrices
from oracle.surfel_oracle import OracleSettings, bin_instances, preprocess
H,W,N=680,1200,200000
raw=make_room_scene(N,"office0",seed=0)
a=activate(raw)
for v in range(8):
    c2w,K=make_camera(0,H,W,mirror=v)
    cm=camera_matrices(c2w[None],K[None],0.001,10.0)
    S=OracleSettings(H,W,cm["tanfov"][0,0].item(),cm["tanfov"][0,1].item(),torch.zeros(4),1.0,cm["viewmatrix"][0],cm["projmatrix"][0])
    with torch.no_grad():
        G=preprocess(a["means"],torch.zeros(N,3),a["opacities"][:,None],a["confidences"],a["colors"],a["scales"],a["rotations"],S)
        so,ranges=bin_instances(G)
    L=(ranges[:,1]-ranges[:,0]).numpy()
    vis=int((G["radii"]>0).sum()) if "radii" in G else -1
    print(v, "visible",vis,"instances(rect)",L.sum(),"max tile",L.max())
