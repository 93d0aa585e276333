"""Import alias: the package sources live in ``rc-mvsnet_amd/`` (a directory name Python's
``import`` statement cannot spell); this shim makes them importable as ``rc_mvsnet_amd``."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rc-mvsnet_amd"))
