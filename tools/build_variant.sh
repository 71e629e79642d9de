#!/bin/bash
# build_variant.sh NAME -DFOO=1 ... : an A/B build of the engine into variants/libzke_NAME.so (use with ZKE_LIB=...)
name=$1; shift; mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZKE_BUILD "$@" -I include -I zkemail.rs_amd/csrc -Wno-unused-function -o variants/libzke_$name.so zkemail.rs_amd/csrc/engine.hip
