"""Compiled ingestion of WhatsHap's own ReadSet / Pedigree objects (whamd_ingest.pyx, built by build.py where the
reference tree is present).  ``load()`` returns the extension module or None -- callers fall back to walking the objects
through their public Python API (whatshap_amd/core.py), which gives the same arrays, only slower."""
import importlib
import sys

_module = None
_tried = False


def load():
    """The compiled module, or None if it was not built / ``whatshap.core`` is not loaded (its C++ symbols must be
    visible: WhatsHap imports it RTLD_GLOBAL, whatshap/__init__.py:6-18)."""
    global _module, _tried
    if _module is not None or _tried:
        return _module
    if "whatshap.core" not in sys.modules:
        return None   # not final: try again once WhatsHap is imported
    _tried = True
    try:
        _module = importlib.import_module("whatshap_amd.ingest.whamd_ingest")
    except ImportError:
        _module = None
        return None
    # The extension reads the C++ objects of whatshap.core through thisptr with the class layouts of the headers it was built
    # against (build.py records that WhatsHap version): refuse to walk the objects of a different WhatsHap.
    import os

    built_for = None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "whamd_ingest.built_for")) as f:
            built_for = f.read().strip()
    except OSError:
        pass
    installed = getattr(sys.modules.get("whatshap"), "__version__", None)
    if built_for not in (None, "unknown") and _is_release(installed) and not _same_release(built_for, installed):
        import warnings

        warnings.warn(f"whatshap_amd.ingest was built against WhatsHap {built_for} but {installed} is loaded: using the objects' "
                      "public Python API instead of the compiled walk", RuntimeWarning)
        _module = None
    return _module


def _is_release(v) -> bool:
    """'2.8', '2.8.1', '2.9.dev3+g1234' -- not the '0+oracle' / '0.1.dev...' stubs of builds without version metadata."""
    import re

    return bool(v) and re.match(r"[1-9]\d*\.\d+", v) is not None


def _same_release(a: str, b: str) -> bool:
    """Versions agree on major.minor."""
    def key(v):
        return tuple(v.replace("+", ".").split(".")[:2])
    return key(a) == key(b)
