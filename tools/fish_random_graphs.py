#!/usr/bin/env python3
"""Runs the random-scene-graph parity tests of tests/test_random_graphs.py over seeds outside the committed set, to look
for scene shapes the committed seeds do not reach (hits, same-seed renders in both shading modes, light lists).   usage: tools/fish_random_graphs.py FIRST LAST   (needs an MI355X)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
first, last = int(sys.argv[1]), int(sys.argv[2])
os.environ["CHIP_TEST_GRAPH_SEEDS"] = "%d:%d" % (first, last)
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_random_graphs.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "300", "--timeout-method", "thread"]))
