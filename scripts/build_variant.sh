#!/bin/bash
# build_variant.sh NAME "-DFOO=1 ..." : builds pcr_grid_search.hip with the given defines and links scripts/bin/libpcr_NAME.so
# from it and the other objects of the in-tree build (A/B runs: PCR_LIB_PATH=scripts/bin/libpcr_NAME.so).
set -e
cd "$(dirname "$0")/../point-cloud-process_amd/csrc"
mkdir -p ../../scripts/bin build/var_$1
/opt/rocm/bin/hipcc $2 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w -I../../include -I. -c pcr_grid_search.hip -o build/var_$1/pcr_grid_search.o
objs=$(ls build/*.o | grep -v pcr_grid_search.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/bin/libpcr_$1.so $objs build/var_$1/pcr_grid_search.o
echo built scripts/bin/libpcr_$1.so
