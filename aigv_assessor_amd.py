"""Import alias: ``import aigv_assessor_amd`` loads the package that lives in ``aigv-assessor_amd/``.

The repo layout mandates the directory name ``aigv-assessor_amd`` (hyphen), which the ``import``
statement cannot spell; this module replaces itself in ``sys.modules`` with that package.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "aigv-assessor_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
