"""Import shim: the package directory is named ``vplines-slam_amd`` (not a valid
Python identifier), so this module turns itself into that package."""
import os as _os

_here = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "vplines-slam_amd")
__path__ = [_here]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(_here, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
