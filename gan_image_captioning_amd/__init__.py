"""Import alias: the product package lives in ``gan-image-captioning_amd/`` (not a
valid Python identifier), this shim makes it importable as
``gan_image_captioning_amd`` by pointing the package search path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gan-image-captioning_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
