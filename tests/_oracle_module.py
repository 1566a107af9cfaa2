"""TEST-ONLY adapter exposing the CPU oracle under the rasterizer module interface
(GaussianRasterizationSettings / GaussianRasterizer), plus a torch-free Adam with the
FusedAdam interface.  Injected into the host-side mirrors by the CPU tests."""
import torch

from oracle.surfel_oracle import OracleSettings, adam_step, rasterize


class GaussianRasterizationSettings:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class GaussianRasterizer:
    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, means3D, means2D, opacities, confidences, shs, colors_precomp, scales, rotations, cov3D_precomp):
        s = self.s
        S = OracleSettings(s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.bg, s.scale_modifier, s.viewmatrix,
                           s.projmatrix, s.sh_degree, s.campos, s.prefiltered, s.render_mask, s.weight_thres, s.debug,
                           s.config)
        return rasterize(means3D, means2D, opacities, confidences, colors_precomp, scales, rotations, S)


class OracleAdam:
    def __init__(self, params, lrs, eps=1e-15):
        self.params, self.lrs, self.eps = list(params), list(lrs), eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        with torch.no_grad():
            adam_step(self.params, [g.reshape(p.shape) for g, p in zip(grads, self.params)], self.m, self.v, self.lrs,
                      self.t, eps=self.eps)
