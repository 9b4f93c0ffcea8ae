"""Import shim: the product package lives in `rosettafold-pytorch_amd/` (the directory name the
build contract fixes; a hyphen is not importable), this makes it importable as
`rosettafold_pytorch_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rosettafold-pytorch_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
