"""Import alias: the package directory is named ``marl-demandresponse-original_amd`` (not a valid Python
identifier), so ``import mdr_amd`` points this package's search path at it and runs its ``__init__``."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "marl-demandresponse-original_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
