"""Host-side cProfile of the mapper loop (config 3 through GaussianMap.update): where the interpreter's time goes.
usage: python profiles/experiments/mapper_cprofile.py [top=45]"""
import cProfile, os, pstats, sys
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
from active_gs_amd.synthetic import make_keyframes, run_mapper_loop
dev = torch.device("cuda:0")
frames = make_keyframes(50, 512, 512, dev)
run_mapper_loop(frames, warmup_frames=2)          # everything loaded
np.random.seed(0)
pr = cProfile.Profile()
pr.enable()
out = run_mapper_loop(frames, warmup_frames=0)
pr.disable()
print(out["seconds"])
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
