#!/bin/bash
# Diagnostic: builds timing-only variants of libsgfhe_hip.so with parts of k_extprod removed
# (results are wrong; only the kernel time matters).  Run the printed commands on the GPU box.
set -e
cd "$(dirname "$0")/../sgfhe.jl_amd/csrc"
mkdir -p ../../tools/abl
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -o ../../tools/abl/lib_$1.so engine.hip 2>/dev/null; }
build base ""
build nobar "-DSGFHE_ABL_NO_BARRIER"
build onebar "-DSGFHE_ABL_ONEBAR"
build nolds "-DSGFHE_ABL_NO_LDS"
build noldsbar "-DSGFHE_ABL_NO_LDS -DSGFHE_ABL_NO_BARRIER"
build notw "-DSGFHE_ABL_NO_TW"
build nokey "-DSGFHE_ABL_NO_KEY"
build nodig "-DSGFHE_ABL_NO_DIG"
build nomem "-DSGFHE_ABL_NO_TW -DSGFHE_ABL_NO_KEY -DSGFHE_ABL_NO_DIG"
build valuonly "-DSGFHE_ABL_NO_TW -DSGFHE_ABL_NO_KEY -DSGFHE_ABL_NO_DIG -DSGFHE_ABL_NO_LDS -DSGFHE_ABL_NO_BARRIER"
ls ../../tools/abl
