"""A/B between two builds of the C-ABI library in one checkout: runs a script with smsut_amd bound to ANOTHER libsmsut_hip.so (e.g. one
compiled with -DSMSUT_F16_X32=0 into scratch/lib_x16/).  usage: python scratch/run_with_lib.py <lib.so> <script.py> [args ...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smsut_amd._hip as H  # noqa: E402

H.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
runpy.run_path(sys.argv[0], run_name="__main__")
