"""Import shim: exposes the directory ``vall-e_amd/`` as the package ``vall_e_amd``.

A hyphen cannot appear in a Python module name, so this module turns itself into
a package whose ``__path__`` is that directory and then runs its ``__init__``.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_here, "vall-e_amd")]
with open(_os.path.join(__path__[0], "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
