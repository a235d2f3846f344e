#!/bin/bash
# tools/tuning/build_variants.sh NAME "FLAGS" ... : kernel-tuning builds of k_iso_adj.hip linked against the current objects
set -e
cd /root/repo/smoothsde_amd/csrc
mkdir -p ../../build/variants
OBJS=$(ls ../../build/obj/*.o | grep -v k_iso_adj.o)
while [ $# -gt 1 ]; do
  NAME=$1; FL=$2; shift 2
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $FL -c k_iso_adj.hip -o ../../build/variants/k_iso_adj_$NAME.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libssde_$NAME.so $OBJS ../../build/variants/k_iso_adj_$NAME.o -ldl && echo built $NAME ) &
done
wait
