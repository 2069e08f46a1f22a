"""Import shim: the package directory is ``revers-o_amd/`` (the name the build
contract fixes), which is not a valid Python identifier.  ``import reverso_amd``
loads that directory as the package ``reverso_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "revers-o_amd")
_spec = importlib.util.spec_from_file_location(
    "reverso_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["reverso_amd"] = _mod
_spec.loader.exec_module(_mod)
