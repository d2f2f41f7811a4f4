#!/usr/bin/env python3
"""A/B tool: run a script of this repo against another build of the library (e.g. the previous commit's, kept as
pyopenvino_amd/libpvhip_prev.so):  python scripts/with_lib.py pyopenvino_amd/libpvhip_prev.so bench.py --steps 20 ..."""
import os, runpy, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device
device.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name='__main__')
