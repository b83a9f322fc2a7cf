#!/usr/bin/env python3
"""Probe the machine for the reference's physics engine.

The reference's physics is the third-party `pybullet` module (`plen_bullet/src/plen_bullet/plen_env.py:6`), not vendored
and not pinned.  This script records, on whatever machine it runs on (build container or GPU box), whether any
importable physics engine or Bullet library exists, so that "PyBullet unavailable" is a measured statement rather than
an assumption.  Output: one JSON object on stdout and in `gpurun_out/pybullet_probe.json` (copied to profiles/).
"""
import importlib
import json
import os
import platform
import subprocess
import sys


def probe():
    mods = {}
    for name in ("pybullet", "pybullet_data", "pybullet_envs", "gym", "gymnasium", "mujoco", "mujoco_py", "pinocchio", "dart", "ode", "raisimpy", "brax"):
        try:
            m = importlib.import_module(name)
            mods[name] = {"importable": True, "version": str(getattr(m, "__version__", "?")), "file": getattr(m, "__file__", None)}
        except Exception as ex:                                   # ModuleNotFoundError normally
            mods[name] = {"importable": False, "error": type(ex).__name__ + ": " + str(ex)}
    try:
        pip = subprocess.run([sys.executable, "-m", "pip", "list", "--format=freeze"], capture_output=True, text=True, timeout=120).stdout
        pip_hits = [l for l in pip.splitlines() if any(k in l.lower() for k in ("bullet", "gym", "mujoco", "physx", "dart"))]
    except Exception as ex:
        pip_hits = ["pip list failed: %r" % (ex,)]
    # any Bullet shared library, header or wheel on disk (bounded search; /proc and /sys skipped)
    hits = []
    try:
        out = subprocess.run("find / -xdev \\( -path /proc -o -path /sys -o -path /dev \\) -prune -o "
                             "\\( -iname '*pybullet*' -o -iname 'libBullet*' -o -iname 'btMultiBody*' -o -iname 'bullet3*' \\) -print 2>/dev/null | head -40",
                             shell=True, capture_output=True, text=True, timeout=300).stdout
        repo = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        hits = [l for l in out.splitlines() if l and not os.path.realpath(l).startswith(repo) and "/root/repo" not in l]
    except Exception as ex:
        hits = ["find failed: %r" % (ex,)]
    # can pip reach an index or a local wheelhouse that has it?  (no install is attempted: --dry-run only)
    try:
        r = subprocess.run([sys.executable, "-m", "pip", "download", "--no-deps", "-d", "/tmp/_probe_wheels", "pybullet"],
                           capture_output=True, text=True, timeout=120)
        pip_dl = {"rc": r.returncode, "tail": (r.stdout + r.stderr)[-300:]}
    except Exception as ex:
        pip_dl = {"rc": None, "tail": repr(ex)}
    return {"host": platform.node(), "python": sys.version.split()[0], "modules": mods, "pip_matches": pip_hits,
            "files_on_disk": hits, "pip_download_pybullet": pip_dl,
            "pybullet_available": bool(mods["pybullet"]["importable"])}


if __name__ == "__main__":
    res = probe()
    txt = json.dumps(res, indent=1)
    print(txt)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pybullet_probe.json"), "w") as f:
        f.write(txt + "\n")
