#!/usr/bin/env python3
"""Development helper: run bench.py against an alternative build of the library (EV2G_LIB=path)."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2gym_amd import engine
engine._LIB_PATH = os.environ["EV2G_LIB"]
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
