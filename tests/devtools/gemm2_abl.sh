#!/bin/bash
# developer tool: where does the persistent GEMM spend its time?  Builds variants of the engine with parts of dgemm2.hip switched
# off (G2_ABL: 1 no C stores, 2 no DMA, 4 no matrix instructions, 8 no drain before the stores; results are WRONG by design) and
# times the two n^3 products of the C2 assembly with each.  usage (container): gemm2_abl.sh build ; (GPU box): gemm2_abl.sh run
cd "$(dirname "$0")/../../scip-sdp_amd" || exit 1
if [ "$1" = build ]; then
   mkdir -p lib/abl
   for v in 0 3; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Icsrc -I../include -Wno-unused-result -DG2_ABL=$v \
         -c csrc/dgemm2.hip -o /tmp/dgemm2_abl$v.o || exit 1
      objs=$(ls build/csrc/*.o | grep -v dgemm2.o)
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/dgemm2_abl$v.o -o lib/abl/libhipsdp_abl$v.so -L/opt/rocm/lib -lrccl \
         -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib || exit 1
   done
   exit 0
fi
for v in 0 3; do
   echo "== G2_ABL=$v"
   HIPSDP_LIB=$PWD/lib/abl/libhipsdp_abl$v.so timeout -k 10 120 python3 ../tests/devtools/gemm2_abl.py
   HIPSDP_GEMM2_SKIP=0 HIPSDP_LIB=$PWD/lib/abl/libhipsdp_abl$v.so timeout -k 10 120 python3 ../tests/devtools/gemm2_abl.py
done
