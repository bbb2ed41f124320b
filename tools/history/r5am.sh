cd $GRAFT_REPO_ROOT
O=gpurun_out/r5am; mkdir -p $O
timeout 400 python tools/class_map.py > $O/map1.log 2>&1; echo rc $?
timeout 400 python tools/class_map.py > $O/map2.log 2>&1; echo rc $?
grep -E "^c[0-9]+ [.o#X]|best|free" $O/map1.log $O/map2.log | cut -c1-300
