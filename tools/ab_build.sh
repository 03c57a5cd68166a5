# builds an A/B variant of the library: tools/ab_build.sh <name> "<extra hipcc flags for decoder3.hip>"  -> tacotron_amd/libtaco_<name>.so
set -e
cd "$(dirname "$0")/../tacotron_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden"
hipcc $FLAGS $2 -c decoder3.hip -o ../../build/obj/decoder3_$1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtaco_$1.so ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder,decoder3_$1,highway,layout,model}.o
echo built libtaco_$1.so
