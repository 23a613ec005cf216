"""Which code a profile belongs to: a content hash of the kernel / host sources.

`.git/` does not travel to the GPU box, so a profile summary records (a) the git hash handed to the collection
script and (b) this content hash; `bench.py` reports the profile's numbers only while (b) still matches the
sources it runs on."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc"))
_EXT = (".hip", ".h", ".cpp", ".inc")


def source_sha(csrc=CSRC):
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith(_EXT) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
