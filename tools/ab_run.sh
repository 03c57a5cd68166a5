# A/B on ONE box: tools/ab_run.sh name1 name2 ...  (each name = tacotron_amd/libtaco_<name>.so; "hip" = the product build); 3 alternating rounds
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for n in "$@"; do echo -n "$n: "; TACO_LIB=$PWD/tacotron_amd/libtaco_$n.so python tools/dec_quick.py --time-only 2>&1 | grep "^S1"; done; done
