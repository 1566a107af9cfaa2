#!/usr/bin/env python3
"""Generates tests/golden/class_surface.json.  Run ONLY in the build container (needs /root/reference).

Imports the reference's host-side Python with its absent third-party imports stubbed (like make_golden.py) and records
the PUBLIC SURFACE of the two classes the drop-in replaces by name - mapping.gaussian_map.GaussianMap and
utils.operations.GaussianRenderer: method names with their parameter names and defaults, property names, and the instance
attributes their constructors set.  Names only - no source text.  tests/test_cpu_host_logic.py holds
active_gs_amd.gaussian_map.GaussianMap / facade.SurfelRenderer to it."""
import inspect
import json
import os
import sys
from types import SimpleNamespace as NS

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

G.install_reference()
import torch  # noqa: E402
from mapping.gaussian_map import GaussianMap  # noqa: E402
from utils.operations import GaussianRenderer  # noqa: E402


def surface(cls):
    methods, props = {}, []
    for name, member in inspect.getmembers(cls):
        if name.startswith("__") and name != "__init__":
            continue
        if isinstance(inspect.getattr_static(cls, name), property):
            props.append(name)
        elif inspect.isfunction(member):
            sig = inspect.signature(member)
            methods[name] = [dict(name=p.name, has_default=p.default is not inspect.Parameter.empty,
                                  default=None if p.default is inspect.Parameter.empty else repr(p.default))
                             for p in sig.parameters.values()]
    return dict(methods=methods, properties=sorted(props))


cfg = NS(bound=[0.001, 10.0], background=[0.0, 0.0, 0.0, 0.0], sparse_ratio=0.1, error_thres=0.25, scale_factor=0.01,
         optimization_steps=10, prune_interval=5, use_view_distribution=True,
         sampler=NS(sampler_type="weighted", batch_size=8, active_size=3),
         optimizer=NS(mean_lr=5e-4, rotation_lr=5e-4, opacity_lr=1e-2, scale_lr=1e-2, harmonic_lr=1e-4))
gm = GaussianMap(cfg, "cpu")
out = dict(GaussianMap=dict(surface(GaussianMap), instance_attributes=sorted(vars(gm))),
           GaussianRenderer=surface(GaussianRenderer))
extr = torch.eye(4)[None]
K = torch.tensor([[[0.866, 0, 0.5], [0, 0.866, 0.5], [0, 0, 1.0]]])
attr = (torch.zeros(1, 3), torch.zeros(1, 1, 3), torch.zeros(1), torch.zeros(1), torch.zeros(1, 3), torch.tensor([[1.0, 0, 0, 0]]))
r = GaussianRenderer(extr, K, attr, torch.zeros(4), (0.001, 10.0), (16, 16), "cpu")
out["GaussianRenderer"]["instance_attributes"] = sorted(vars(r))
json.dump(out, open(os.path.join(HERE, "class_surface.json"), "w"), indent=1)
print({k: (len(v["methods"]), len(v["properties"]), len(v["instance_attributes"])) for k, v in out.items()})
