"""Import alias: ``import unet_convlstm_amd`` loads the package in ``unet-convlstm_amd/``."""
import importlib
import sys

_pkg = importlib.import_module("unet-convlstm_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("unet-convlstm_amd."):
        sys.modules["unet_convlstm_amd." + _name.split(".", 1)[1]] = _mod
sys.modules[__name__] = _pkg
