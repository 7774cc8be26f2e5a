from spacefortress_amd.env import SSF_Env  # noqa: F401
