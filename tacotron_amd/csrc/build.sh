#!/bin/bash
# Builds libtaco_hip.so for gfx950 in-tree (tacotron_amd/libtaco_hip.so).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libtaco_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden"
mkdir -p ../../build/obj
pids=()
# decoder3.hip is compiled with LLVM's "iterative-maxocc" machine-scheduling strategy: its step loops are issue-bound chains of
# ~4,000 instructions per wave and the schedule decides how many wait states they carry.  Measured on the same box, three
# alternations (profiles/r05_dec_sched_ab.txt): BPTT 14.19 -> 13.66 us per decoder step, forward unchanged, S1 step -0.1 ms;
# max-ilp / max-memory-clause / iterative-minreg are slower, the strategy does nothing for bigru.hip.
D3FLAGS="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
for f in gemm gemm2 vocoder elementwise bigru decoder highway prenet layout model; do
  hipcc $FLAGS -c $f.hip -o ../../build/obj/$f.o &
  pids+=($!)
done
hipcc $FLAGS $D3FLAGS -c decoder3.hip -o ../../build/obj/decoder3.o &
pids+=($!)
# probe build of the decoder (timing probes compiled in; tools/dec_probe.py loads it through TACO_LIB)
hipcc $FLAGS -DTACO_DEC_PROBES -c decoder.hip -o ../../build/obj/decoder_probe.o &
pids+=($!)
hipcc $FLAGS $D3FLAGS -DTACO_DEC_PROBES -c decoder3.hip -o ../../build/obj/decoder3_probe.o &
pids+=($!)
# the round-3 decoder form (column sums by wave, 8-byte polls, no poll-shadow work, per-lane poll loops, tanh in its sum form; this
# round's compiler flags): bench.py alternates it with the product build on the box it runs on (`ab.decoder`), so that a decoder gain
# or loss is visible on the driver's box and not only on the builder's
hipcc $FLAGS $D3FLAGS -DTACO_NO_RS -DTACO_NO_POLL128 -DTACO_NO_SHADOW -DTACO_NO_GROUPED_FANDQ -DTACO_NO_UNIPOLL -DTACO_NO_TANH_SPLIT -c decoder3.hip -o ../../build/obj/decoder3_prev.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder,decoder3,highway,prenet,layout,model}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtaco_probe.so ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder_probe,decoder3_probe,highway,prenet,layout,model}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtaco_prevdec.so ../../build/obj/{gemm,gemm2,vocoder,elementwise,bigru,decoder,decoder3_prev,highway,prenet,layout,model}.o
echo "built $(realpath $OUT)"
