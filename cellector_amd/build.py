"""Builds libcellector_hip.so (hipcc, gfx950) and the host binary in-tree."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)


def build_library(jobs=8, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return os.path.join(_HERE, "libcellector_hip.so")


def build_host(verbose=False):
    host = os.path.join(ROOT, "host")
    if not os.path.exists(os.path.join(host, "Makefile")):
        return None
    cmd = ["make", "-C", host]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return os.path.join(host, "cellector")
