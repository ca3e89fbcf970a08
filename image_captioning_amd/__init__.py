"""Import alias: the product package lives in the directory `image-captioning_amd/` (a hyphen is not
importable), so `import image_captioning_amd` resolves its submodules there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "image-captioning_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
