for i in 1 2 3 4 5 6; do timeout 120 python tools/boundary_once.py 1.0 5 2>&1 | grep "^call" | awk '{printf "%s ", $3}'; echo; done
timeout 300 python tools/gpu_boundary.py 1.0 60 2>&1 | grep -E "pinned packed" | cut -c1-700
