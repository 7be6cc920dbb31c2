"""The switch table of stylemesh_amd/runtime/config.py is complete and has no dead entries."""
import os
import re

from conftest import REPO


def _read_names():
    names = set()
    roots = [os.path.join(REPO, "stylemesh_amd"), os.path.join(REPO, "bench.py")]
    for root in roots:
        files = [root] if os.path.isfile(root) else [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs
                                                     if f.endswith((".py", ".hip", ".h"))]
        for path in files:
            if path.endswith(os.path.join("runtime", "config.py")):
                continue
            text = open(path, errors="ignore").read()
            for m in re.finditer(r"(?:environ(?:\.get)?\s*[\[(]\s*|getenv\(\s*)\"((?:STYLEMESH|SM)_[A-Z0-9_]+)\"", text):
                names.add(m.group(1))
    return names


def test_every_switch_is_listed_and_every_listed_switch_is_read():
    from stylemesh_amd.runtime.config import SWITCHES
    read = _read_names()
    assert len(read) >= 25
    missing = sorted(read - set(SWITCHES))
    dead = sorted(set(SWITCHES) - read)
    assert not missing, f"read but not in runtime/config.py: {missing}"
    assert not dead, f"listed in runtime/config.py but read nowhere: {dead}"
    for name, (default, kind, what) in SWITCHES.items():
        assert kind in ("mode", "tuning", "experiment", "diagnostic") and len(what) > 20, name
