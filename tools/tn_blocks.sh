#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1024 1536 2048 3072 4096; do
TACO_TN_BLOCKS=$b python tools/family_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fam_b$b.txt
echo "TACO_TN_BLOCKS=$b: $(grep -E '^step' gpurun_out/fam_b$b.txt) $(grep -E '^sum' gpurun_out/fam_b$b.txt)  tn sum: $(grep ' tn' gpurun_out/fam_b$b.txt | awk '{s+=$2} END {print s}')"
done
paste <(grep " tn" gpurun_out/fam_b1024.txt | cut -c1-16) <(grep " tn" gpurun_out/fam_b1536.txt | cut -c5-16) <(grep " tn" gpurun_out/fam_b2048.txt | cut -c5-16) <(grep " tn" gpurun_out/fam_b3072.txt | cut -c5-16) <(grep " tn" gpurun_out/fam_b4096.txt | cut -c5-100)
