cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/dec_quick.py --time-only 2>&1 | grep "^S1"; }
run A=1
run TACO_GEMM2_MIN_TILES=96
run TACO_GEMM2_MIN_TILES=64
run TACO_GEMM2_MIN_TILES=32
run TACO_TN_BLOCKS=1536
run TACO_TN_BLOCKS=6144
run TACO_NO_OVERLAP=1
