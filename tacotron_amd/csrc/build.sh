#!/bin/bash
# Builds libtaco_hip.so for gfx950 in-tree (tacotron_amd/libtaco_hip.so).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libtaco_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden"
mkdir -p ../../build/obj
pids=()
for f in gemm gemm2 vocoder elementwise bigru decoder decoder3 highway prenet layout model; do
  hipcc $FLAGS -c $f.hip -o ../../build/obj/$f.o &
  pids+=($!)
done
# probe build of the decoder (timing probes compiled in; tools/dec_probe.py loads it through TACO_LIB)
hipcc $FLAGS -DTACO_DEC_PROBES -c decoder.hip -o ../../build/obj/decoder_probe.o &
pids+=($!)
hipcc $FLAGS -DTACO_DEC_PROBES -c decoder3.hip -o ../../build/obj/decoder3_probe.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder,decoder3,highway,prenet,layout,model}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtaco_probe.so ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder_probe,decoder3_probe,highway,prenet,layout,model}.o
echo "built $(realpath $OUT)"
