#!/bin/bash
# r6: the full GPU suite as the driver runs it, the graph-gap probe, then the round's profiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 120 tools/ubench/graph_gap > $O/graph_gap.txt 2>&1; cat $O/graph_gap.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash profiles/r06_scripts/r06_profile.sh > $O/profile.log 2>&1; tail -40 $O/profile.log
