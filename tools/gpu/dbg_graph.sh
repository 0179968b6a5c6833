python tools/gpu/dbg_graph.py default 2>&1 | tail -8
NBE_FUSE=0 python tools/gpu/dbg_graph.py nofuse 2>&1 | tail -8
