#!/bin/bash
# A/B builds of one kernel file: profiles/build_variant.sh NAME FILE.hip "-DFLAG ..." links a copy of the
# library with FILE.hip compiled under the extra flags into profiles/variants/lib_NAME.so; run a
# bench or microbench against it with RX_LIB_PATH=profiles/variants/lib_NAME.so.
set -e
cd "$(dirname "$0")/../recsys_amd/csrc"
name=$1; file=$2; flags=$3
mkdir -p ../../profiles/variants
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $flags -c "$file" -o "/tmp/variant_${name}.o"
objs=""
for f in rx_gather rx_interact rx_sort rx_update rx_fields rx_gemm rx_norm rx_eval rx_sample rx_neumf rx_blaslt; do
  if [ "$f.hip" == "$file" ]; then objs="$objs /tmp/variant_${name}.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -L/opt/rocm/lib -lhipblaslt -o "../../profiles/variants/lib_${name}.so"
echo "profiles/variants/lib_${name}.so"
