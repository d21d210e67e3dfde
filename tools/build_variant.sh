#!/bin/bash
# Scratch: builds libpt_hip.so with extra compiler flags into build/var_<name>/ (next to a copy of libpt_host.so), for
# A/B runs with PT_HIP_LIB_DIR=build/var_<name>.   Usage: tools/build_variant.sh <name> [flags...]
# (only pt_hip.hip is recompiled; the tree builder's object is taken from the regular build)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -munsafe-fp-atomics -Wno-unused-function \
    -Iinclude -Ibuild "$@" -c -o build/var_$name/pt_hip.o path-tracer_amd/csrc/pt_hip.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/var_$name/libpt_hip.so build/var_$name/pt_hip.o build/obj/bvh_build.o
rm -f build/var_$name/pt_hip.o
cp path-tracer_amd/lib/libpt_host.so build/var_$name/
