#!/bin/bash
# whole GPU suite in one process, the smoke entry, then the round's measurement set (tag $1)
TAG=${1:-r02s}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/${TAG}_gpu_tests_all.log 2>&1; rc=$?
tail -4 gpurun_out/${TAG}_gpu_tests_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
bash tools/scripts/r02_profile.sh $TAG
