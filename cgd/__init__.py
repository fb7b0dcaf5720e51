"""Drop-in import path of the reference (`from cgd.cgd import clip_guided_diffusion`, `from cgd import losses, ...`).
The modules live in clip-guided-diffusion_amd/cgd/ (the MI355X host-side mirror of the reference interface)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
import cgd_amd as _cgd_amd  # noqa: E402,F401  (registers the hyphenated package directory)

__path__.insert(0, _os.path.join(_root, "clip-guided-diffusion_amd", "cgd"))
