# builds an A/B variant of the library: tools/ab_build.sh <name> "<extra hipcc flags>" [source, default decoder3]
#   -> tacotron_amd/libtaco_<name>.so with <source>.hip compiled with the extra flags, every other object from build/obj
set -e
cd "$(dirname "$0")/../tacotron_amd/csrc"
SRC=${3:-decoder3}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden"
hipcc $FLAGS $2 -c $SRC.hip -o ../../build/obj/${SRC}_$1.o
OBJS=""
for o in gemm gemm2 vocoder elementwise bigru decoder decoder3 highway prenet layout model; do
  if [ "$o" = "$SRC" ]; then OBJS="$OBJS ../../build/obj/${SRC}_$1.o"; else OBJS="$OBJS ../../build/obj/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtaco_$1.so $OBJS
echo built libtaco_$1.so
