#!/bin/bash
# round 4, final job: the whole GPU suite (both DLT forms), smoke, the profile sets and the bench lines of the final build
cd /root/repo; O=gpurun_out/r4_final; mkdir -p $O; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log ); tail -n 6 $O/pytest.log
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 ); tail -n 2 $O/smoke.log
rm -f gpurun_out/pmc_traffic.json
bash tools/r4_profiles.sh 2>&1 | grep -E "^gpurun_out/r04_final|pass .* rc=" 
