"""CPU oracle of the OMG-Planner CHOMP path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (omg-planner_amd/) never does.
"""
from .oracle import *  # noqa: F401,F403
