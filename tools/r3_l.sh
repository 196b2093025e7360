cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 1500 python3 tests/stress_parity.py 6000 3031 2>&1 | tail -4
timeout 600 python3 tests/stress_points.py 2000 3032 2>&1 | tail -3
timeout 600 python3 tests/stress_window.py 300 3033 2>&1 | tail -3
