"""Module name the reference imports (/root/reference/utils/operations.py:22-25).

Putting this repository's root on ``sys.path`` makes ActiveGS's ``utils.operations`` pick
up the MI355X rasterizer with no change to the reference."""
from active_gs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, check_overflow

# check_overflow(): optional - waits for the deferred workspace checks of the forward passes issued so far (see
# active_gs_amd/rasterizer.py: STATUS_CHECK); the two names above are all the reference uses.
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "check_overflow"]
