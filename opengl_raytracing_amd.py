"""Import shim: the package directory is `opengl-raytracing_amd/` (not a valid identifier), so
`import opengl_raytracing_amd` loads it from there under this name."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "opengl-raytracing_amd"
_spec = importlib.util.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
