"""A fingerprint of the device sources: profiles carry it, and bench.py only quotes a profile's counters while the
fingerprint still matches the sources the library in use was built from (a counter replayed after the kernel changed
says nothing)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_sources_stamp():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + glob.glob(os.path.join(_HERE, "csrc", "Makefile")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
