"""Builds libbenerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=4):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j", str(jobs)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libbenerf_hip.so failed")
    lib = os.path.join(HERE, "libbenerf_hip.so")
    assert os.path.exists(lib), lib
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
