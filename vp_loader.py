"""Import helper: the package directory is named ``virgo-plus_amd`` (hyphen), so it cannot be imported
with a plain ``import``.  ``load()`` registers it as the module ``virgo_plus_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "virgo_plus_amd" in sys.modules:
        return sys.modules["virgo_plus_amd"]
    pkg = os.path.join(ROOT, "virgo-plus_amd")
    spec = importlib.util.spec_from_file_location("virgo_plus_amd", os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["virgo_plus_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
