#!/usr/bin/env python3
"""Run tools/exp_variants.py against an experimental build: python tools/exp_lib.py LIB.so [exp_variants args...]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pim-compression_amd"))
import snappy_hip_binding as shb
shb.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "tools", "exp_variants.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
