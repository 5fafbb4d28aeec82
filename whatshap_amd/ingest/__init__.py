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
    return _module
