#!/bin/bash
# Random-scene-graph parity sweeps of this round (see profiles/r03/seed_sweep.txt): the committed tests over seeds outside
# the committed set, default settings, the variant with discs of any normal and media in turned boxes, and stress settings.
run() { echo "== $1"; shift; env "$@" timeout -k 10 500 python tools/fish_random_graphs.py 0 1000 2>&1 | tail -1; }
run "default, seeds 0-1000"
run "variant 1 (discs of any normal, media bounded by turned boxes), seeds 0-1000" CHIP_TEST_GRAPH_VARIANT=1
run "variant 2 (media whose boundaries hold media of their own), seeds 0-1000" CHIP_TEST_GRAPH_VARIANT=2
run "CHIP_SMALL_STACK=4 (most descents of the compact form overflow to the literal pass)" CHIP_SMALL_STACK=4
run "CHIP_LDS_STACK=2 CHIP_NO_LDS_CACHE=1" CHIP_LDS_STACK=2 CHIP_NO_LDS_CACHE=1
run "CHIP_POOL_PATHS=8192 CHIP_STREAMS=3 CHIP_LDS_STACK=3" CHIP_POOL_PATHS=8192 CHIP_STREAMS=3 CHIP_LDS_STACK=3
run "CHIP_NO_FUSED_FIRST=1 (bounce 0 in two kernels)" CHIP_NO_FUSED_FIRST=1
run "CHIP_NO_SHADE_TABLES=1 (record tables left in memory for the shade kernels)" CHIP_NO_SHADE_TABLES=1
run "CHIP_TAIL_PATHS=0 (every bounce in its own launches)" CHIP_TAIL_PATHS=0
run "CHIP_TAIL_PATHS=2000 CHIP_SMALL_STACK=4 (the tail kernel takes over late; most of its rays go through the literal form inside it)" CHIP_TAIL_PATHS=2000 CHIP_SMALL_STACK=4
