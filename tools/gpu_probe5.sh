# round 5: what the write stream of the capture loop costs, by bytes per line and burst length (tools/probe_capture.hip, variant 5)
cd $GRAFT_REPO_ROOT
./build/probe/probe_capture 10000 5
./build/probe/probe_capture 10000 5
