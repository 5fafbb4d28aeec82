#!/usr/bin/env python3
"""Builds whatshap_amd/ingest/whamd_ingest*.so -- the compiled ingestion of WhatsHap's own ReadSet / Pedigree objects.

Needs what any sibling extension of WhatsHap needs: its ``whatshap/core.pxd`` / ``cpp.pxd`` and ``src/*.h`` (the
reference tree, WHATSHAP_REFERENCE, default /root/reference).  Where the tree is absent (the GPU box) the prebuilt
shared object travels with the snapshot and nothing is rebuilt.  Our own recipe: cython + g++, flags of the reference's
setup.py:8-16."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("WHATSHAP_REFERENCE", "/root/reference")


def target():
    return os.path.join(HERE, "whamd_ingest" + sysconfig.get_config_var("EXT_SUFFIX"))


def available():
    return os.path.exists(target())


def build(force=False):
    """Returns True if the shared object exists afterwards."""
    if not os.path.isdir(os.path.join(REFERENCE, "whatshap")):
        return available()
    try:
        import Cython  # noqa: F401
        import numpy  # noqa: F401
    except ImportError:
        return available()
    sources = [os.path.join(HERE, "whamd_ingest.pyx"), os.path.join(HERE, "whamd_ingest_helpers.h")]
    if not force and available() and os.path.getmtime(target()) >= max(os.path.getmtime(p) for p in sources):
        return True
    gen = os.path.join(HERE, "build")
    os.makedirs(gen, exist_ok=True)
    cpp = os.path.join(gen, "whamd_ingest.cpp")
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-I", REFERENCE, sources[0], "-o", cpp],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-pthread", "-w", "-Werror=return-type",
                           "-I" + os.path.join(REFERENCE, "src"), "-I" + HERE, "-I" + sysconfig.get_paths()["include"],
                           "-o", target(), cpp])
    with open(os.path.join(HERE, "whamd_ingest.built_for"), "w") as f:   # read by load(): class layouts are per WhatsHap release
        f.write(reference_version() + "\n")
    return available()


def reference_version():
    """Release of the tree the extension is built against: the newest heading of its CHANGES.rst ('v2.8 (2025-06-08)')."""
    import re

    try:
        with open(os.path.join(REFERENCE, "CHANGES.rst")) as f:
            for line in f:
                m = re.match(r"v(\d+\.\d+)", line)
                if m:
                    return m.group(1)
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    print("compiled ingestion:", "built" if build(force="--force" in sys.argv) else "NOT available")
