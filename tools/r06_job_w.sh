#!/bin/bash
O=gpurun_out/r06w; mkdir -p $O
EAO_BA_WALL_STAMPS=1 python3 tools/dbg_ba_cabi.py 2>&1 | tail -n 6 | cut -c1-200
EAO_BA_SPIN=1 EAO_BA_WALL_STAMPS=1 python3 tools/dbg_ba_cabi.py 2>&1 | tail -n 4 | cut -c1-200
