"""Import shim: the package lives in the directory ``t2v-turbo_amd/`` (the name the project
prescribes, which is not a valid Python identifier); ``import t2v_turbo_amd`` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "t2v-turbo_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
