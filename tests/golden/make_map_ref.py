#!/usr/bin/env python3
"""Generates tests/golden/map_ref.th with the REFERENCE's own ``GaussianMap.save``
(/root/reference/mapping/gaussian_map.py:491-507) and checks, with the reference's own ``GaussianMap.load`` (:509-527),
that a file written by this repository's ``map_io.save_map`` loads there.  Run ONLY in the build container (needs
/root/reference); the fixture holds tensors and scalars, no source text.  tests/test_cpu_host_logic.py reads it."""
import json
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import AttrDict, install_reference, mapper_cfg  # noqa: E402


def main():
    ops, gm, _ = install_reference()
    from active_gs_amd import map_io
    from active_gs_amd.synthetic import make_room_scene
    n = 300
    raw = make_room_scene(n, seed=77)
    g = torch.Generator().manual_seed(78)
    m = gm.GaussianMap(mapper_cfg(2), "cpu")
    m._means, m._scales, m._rotations = raw["means"].clone(), raw["scales"].clone(), raw["rotations"].clone()
    m._opacities, m._harmonics = raw["opacities"].clone(), raw["harmonics"].clone()
    m.view_scores = torch.rand(n, generator=g)
    m.view_supports = torch.randint(0, 5, (n,), generator=g).float()
    m.view_means = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    m.save(HERE, "ref")                                            # -> map_ref.th, by the reference's own code
    st = torch.load(os.path.join(HERE, "map_ref.th"))
    # the other direction: our writer -> the reference's reader
    raw2, cfg = map_io.load_map(os.path.join(HERE, "map_ref.th"))

    class T:                                                      # what map_io.map_state reads off a trainer
        pass
    t = T()
    for k, v in raw2.items():
        setattr(t, k, v)
    t.cfg = dict(bound=cfg["bound"], use_view_distribution=cfg["use_view_distribution"], scale_factor=cfg["scale_factor"])
    t.background = torch.tensor(cfg["background"])
    with tempfile.TemporaryDirectory() as d:
        path = map_io.save_map(t, d, "ours")
        m2 = gm.GaussianMap(mapper_cfg(2), "cpu")
        m2.load(path)                                             # the reference's load(): every key access must work
        ok = all(torch.equal(getattr(m2, a), st[b]) for a, b in (("_means", "means"), ("_scales", "scales"),
                                                                  ("_harmonics", "harmonics"), ("_opacities", "opacities"),
                                                                  ("_rotations", "rotations"), ("view_scores", "view_scores"),
                                                                  ("view_supports", "view_supports"), ("view_means", "view_means")))
        ok = ok and m2.scene_near == st["near"] and m2.scene_far == st["far"] and m2.scale_factor == st["scale_factor"]
        ok = ok and torch.equal(m2.background_color, torch.tensor(st["background_color"], dtype=torch.float32)) and m2.is_init
    json.dump(dict(reference_load_of_map_io_file="ok" if ok else "MISMATCH",
                   keys=sorted(st.keys()), types={k: type(v).__name__ for k, v in st.items()}),
              open(os.path.join(HERE, "map_ref.json"), "w"), indent=1)
    print("map_ref.th written;", "reference load() of a map_io file:", "ok" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
