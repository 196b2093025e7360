cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== tier M"; timeout 300 python3 tools/diag_fast.py 2>&1 | tail -22
echo "== tier L"; MVR_TIER=L timeout 300 python3 tools/diag_fast.py 2>&1 | tail -22
echo "== prepass M"; timeout 300 python3 tools/diag_prepass.py 2>&1 | tail -12
echo "== prepass L"; MVR_TIER=L timeout 300 python3 tools/diag_prepass.py 2>&1 | tail -12
timeout 600 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
