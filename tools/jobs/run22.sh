cd $GRAFT_REPO_ROOT; timeout 600 python tools/jobs/dpcheck.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
