#!/bin/bash
# B=1 and B=32 inference kernel timelines (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for B in 1 32; do
  rm -rf /tmp/inf$B
  rocprofv3 --kernel-trace -d /tmp/inf$B -o inf -- python $R/tools/infer_timeline.py run $B > /tmp/inf$B.log 2>&1
  DB=$(find /tmp/inf$B -name '*.db' | head -1)
  python $R/tools/infer_timeline.py show $DB 4 > $R/gpurun_out/r04_infer_timeline_B$B.txt 2>&1
done
tail -3 $R/gpurun_out/r04_infer_timeline_B1.txt
