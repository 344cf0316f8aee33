"""pytest with smsut_amd bound to another build of the library.  usage: python scratch/pytest_with_lib.py <lib.so> <pytest args ...>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smsut_amd._hip as H  # noqa: E402
import pytest  # noqa: E402

H.LIB_PATH = os.path.abspath(sys.argv[1])
sys.exit(pytest.main(sys.argv[2:]))
