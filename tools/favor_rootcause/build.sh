#!/bin/bash
# Builds the two experiment libraries (NOT part of the product; see README.md in this directory):
#   libfvx0.so  the fused FAVOR+ kernel exactly as it stood at commit 61ce90f (the form that was not run-to-run reproducible)
#   libfvx1.so  the same kernel + checksums of the Q fragment registers (FVX_CHECK=1)
set -e
cd "$(dirname "$0")"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -fno-honor-nans -fno-signed-zeros"
/opt/rocm/bin/hipcc $FLAGS -DFVX_CHECK=0 -shared favor_exp.hip -o libfvx0.so
/opt/rocm/bin/hipcc $FLAGS -DFVX_CHECK=1 -shared favor_exp.hip -o libfvx1.so
ls -la libfvx0.so libfvx1.so
