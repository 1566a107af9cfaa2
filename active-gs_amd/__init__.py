# The importable name of this directory is ``active_gs_amd`` (see ../active_gs_amd.py).
