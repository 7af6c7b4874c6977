"""Import shim: the package sources live in ``multiview-stitcher_amd/`` (the
directory name the project layout prescribes, which is not a valid Python
identifier).  This module makes ``import multiview_stitcher_amd`` resolve to
that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multiview-stitcher_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
