"""Import alias: the package directory name required by the repo layout contains hyphens, which the `import`
statement cannot spell.  `import trimodal_amd` gives the same module object."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
sys.modules[__name__] = _pkg
