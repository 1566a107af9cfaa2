"""Import shim: the package directory is ``active-gs_amd/`` (not a valid Python
identifier), so this module lends it an importable name.  ``import active_gs_amd``
and ``import active_gs_amd.rasterizer`` resolve into that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "active-gs_amd")]
__version__ = "0.1.0"
