"""Import shim: exposes the package directory `text-to-video-finetuning_amd/` as the module `t2v_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text-to-video-finetuning_amd")
_spec = importlib.util.spec_from_file_location("t2v_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["t2v_amd"] = _pkg
_spec.loader.exec_module(_pkg)
