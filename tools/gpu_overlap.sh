cd $GRAFT_REPO_ROOT
timeout 600 python tools/overlap_probe.py 10000 10 2>&1 | tail -4
timeout 600 python tools/overlap_probe.py 2500 20 2>&1 | tail -4
timeout 900 python tools/k1_ab.py 4 build/ab/r05_start.so build/variants/*.so 2>&1 | tail -6
