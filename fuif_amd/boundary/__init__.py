"""C++ boundary layer (source-compatible Image/Channel/Transform + fuif_decode_file); see README there."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    mk = os.path.join(_HERE, "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
