"""puts <repo>/tools on sys.path for the tests that drive tools/*.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
