"""`python -m carskit_amd.main -c setting.conf`: runs the product host, the C++ driver over the C ABI (carskit_amd/csrc/host, built to
carskit_amd/bin/carskit-mi355x by __graft_entry__.build()).  The reference's flow -- CARSKit.java: execute :109, preset :140, readData
:220, runAlgorithm :310, runCrossValidation :388 -- is implemented there; a Python restatement used by the tests to put the CPU oracle
behind the same host logic lives in tests/hostmirror/."""
import argparse
import os
import subprocess
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="carskit_amd")
    ap.add_argument("-c", dest="configs", action="append", default=None)
    args = ap.parse_args(argv)
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "carskit-mi355x")
    if not os.path.exists(exe):
        raise FileNotFoundError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % exe)
    cmd = [exe]
    for c in args.configs or ["setting.conf"]:
        cmd += ["-c", c]
    return subprocess.call(cmd)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
