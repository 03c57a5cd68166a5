#!/bin/bash
# which vendor kernels torch.mm picks for the step's shapes (names encode macro tile / MFMA / depth)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/bn
cat > /tmp/bn.py <<'PY'
import torch
for (M,N,K) in [(4096,4096,4096),(11520,1024,256),(11520,1025,256),(11520,256,1024),(11520,1024,1024),(6400,128,2048),(6400,2048,128),(11520,128,768),(11520,768,128)]:
    A=torch.randn(M,K,device='cuda'); B=torch.randn(K,N,device='cuda')
    for _ in range(3): torch.mm(A,B)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d /tmp/bn -o bn -- python /tmp/bn.py > /tmp/bn.log 2>&1
DB=$(find /tmp/bn -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/r04_blas_names.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
g = [x for x in cols if 'grid' in x and x.endswith('x')][0]; w = [x for x in cols if 'workgroup' in x and x.endswith('x')][0]
lds = [x for x in cols if 'lds' in x.lower()]
extra = (', ' + lds[0]) if lds else ''
for r in c.execute("select name, start, end, %s, %s%s from kernels order by start" % (g, w, extra)):
    if 'Cijk' in r[0] or 'gemm' in r[0].lower():
        print('%8.1f us grid %d wg %d lds %s  %s' % ((r[2]-r[1])/1e3, r[3], r[4], r[5] if lds else '?', r[0][:400]))
PY
