"""CPU, uses the oracle (hence under tests/): how many of a wave's 64 pixels take a surfel per (surfel, 8x8 quadrant)
pair of the bench view - the lane utilisation of the blend loops (63 %: a finer pixel granularity would not pay)."""
import sys, os, math
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from active_gs_amd import env_config
env_config.apply_env(os.environ)   # the package itself reads no environment variable
import active_gs_amd
from active_gs_amd.synthetic import make_camera, make_room_scene, activate
from active_gs_amd.camera import camera_matrices
from oracle.surfel_oracle import OracleSettings, preprocess
N, H, W = 200000, 680, 1200
raw = make_room_scene(N, "office0", seed=0)
a = activate(raw)
c2w, K = make_camera(0, H, W)
cm = camera_matrices(c2w[None], K[None], 0.001, 10.0)
tanx, tany = cm["tanfov"][0, 0].item(), cm["tanfov"][0, 1].item()
S = OracleSettings(H, W, tanx, tany, torch.zeros(4), 1.0, cm["viewmatrix"][0], cm["projmatrix"][0])
with torch.no_grad():
    G = preprocess(a["means"], torch.zeros(N, 3), a["opacities"][:, None], a["confidences"], a["colors"], a["scales"], a["rotations"], S)
m = G["mean2D"].numpy().astype(np.float64); con = G["conic"].numpy().astype(np.float64); op = G["opacity"].numpy().astype(np.float64)
rect = G["rect"].numpy()
print("visible", len(op), "rect instances", int(G["ntiles"].sum()))
# per (surfel, 8x8 quadrant): count pixels with alpha >= 1/255
hist = np.zeros(65, dtype=np.int64)
hist4 = np.zeros(17, dtype=np.int64)  # 4x4 subblocks active within reached quadrants
pairs_tile = 0
px = np.arange(8)[None, :]; py = np.arange(8)[:, None]
for i in range(len(op)):
    x0, y0, x1, y1 = rect[i]
    xs = np.arange(x0 * 16, x1 * 16); ys = np.arange(y0 * 16, y1 * 16)
    dx = m[i, 0] - xs[None, :]; dy = m[i, 1] - ys[:, None]
    power = -0.5 * (con[i, 0] * dx * dx + con[i, 2] * dy * dy) - con[i, 1] * dx * dy
    al = np.minimum(0.99, op[i] * np.exp(power))
    ok = (power <= 0) & (al >= 1 / 255)
    ok &= (xs[None, :] < W) & (ys[:, None] < H)
    h, w = ok.shape
    q = ok.reshape(h // 8, 8, w // 8, 8).sum(axis=(1, 3))
    t = ok.reshape(h // 16, 16, w // 16, 16).sum(axis=(1, 3))
    pairs_tile += int((t > 0).sum())
    hist += np.bincount(q.ravel(), minlength=65)
    s4 = ok.reshape(h // 4, 4, w // 4, 4).sum(axis=(1, 3)) > 0
    s4q = s4.reshape(h // 8, 2, w // 8, 2).sum(axis=(1, 3))
print("tile instances with >=1 px", pairs_tile)
tot = hist[1:].sum()
print("quadrant pairs with >=1 px", tot, "empty quadrant candidates", hist[0])
act = (hist * np.arange(65)).sum()
print("mean active lanes per nonempty pair %.1f" % (act / tot))
cum = np.cumsum(hist[1:]) / tot
for k in (4, 8, 16, 32, 48, 63): print("pairs with <=%d px: %.3f" % (k, cum[k - 1]))
print("full 64: %.3f" % (hist[64] / tot))
