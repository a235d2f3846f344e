#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "FLAGS": rebuild ONE translation unit with extra flags and link it with the
# other objects of the regular build into smoothsde_amd/lib/libssde_hip_NAME.so (kernel-tuning A/B: SSDE_LIB=... selects it)
set -e
cd "$(dirname "$0")/../smoothsde_amd/csrc"
NAME=$1; FILE=$2; FLAGS=$3
OBJ=../../build/obj
mkdir -p $OBJ/variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $FILE -o $OBJ/variants/${FILE%.hip}_$NAME.o
SRCS=$(grep '^SRCS' Makefile | sed 's/SRCS := //')
OBJS=""
for f in $SRCS; do
  if [ "$f" == "$FILE" ]; then OBJS="$OBJS $OBJ/variants/${FILE%.hip}_$NAME.o"; else OBJS="$OBJS $OBJ/${f%.hip}.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libssde_hip_$NAME.so $OBJS -ldl
echo built ../lib/libssde_hip_$NAME.so
