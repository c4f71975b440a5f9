#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -6 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
