export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/ab_run.sh hip mb4 mb8 cb8 > $O/r04g_ab.txt 2>&1
python tools/dec3_trace.py 32 > $O/r04g_dec_probes.txt 2>&1
cat $O/r04g_ab.txt; cat $O/r04g_dec_probes.txt
