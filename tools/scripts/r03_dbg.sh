cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "=== A (previous kernel)"; CASSNAT_HIP_LIB=$GRAFT_REPO_ROOT/ab/libA.so timeout -k 5 200 python tools/chain_err.py 2>&1 | tail -9
echo "=== B (new kernel)"; timeout -k 5 200 python tools/chain_err.py 2>&1 | tail -9
