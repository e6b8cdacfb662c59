"""Import alias: the package directory is ``qat-vit_amd/`` (not a valid Python identifier);
``import qat_vit_amd`` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qat-vit_amd")
_spec = importlib.util.spec_from_file_location("qat_vit_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["qat_vit_amd"] = _mod
_spec.loader.exec_module(_mod)
