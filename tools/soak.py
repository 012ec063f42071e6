"""Launcher of the randomised GPU parity soak (tests/soak.py):  python tools/soak.py [seconds] [first seed] [host threads]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soak

soak.main()
