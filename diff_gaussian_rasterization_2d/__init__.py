"""Module name the reference imports (/root/reference/utils/operations.py:22-25).

Putting this repository's root on ``sys.path`` makes ActiveGS's ``utils.operations`` pick
up the MI355X rasterizer with no change to the reference."""
from active_gs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
