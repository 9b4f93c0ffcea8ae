#!/bin/bash
# Build one bf16 library per compile-time ablation value of csrc/ffn.hip (FFN_ABL bits, see the source) next to the production
# library: rosettafold-pytorch_amd/librfmi_ffnabl<V>.so.  Run on the GPU box with RFMI_LIB=<that file> FFN_QUICK=1 python tools/ffn_bench.py
#   bash tools/ffn_ablation.sh 1 3 8 9 15 31 63
set -e
cd "$(dirname "$0")/../rosettafold-pytorch_amd/csrc"
make -j8 >/dev/null
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -DFFN_ABL=$v -c ffn.hip -o /tmp/ffn_abl_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm.o gemm_fast.o gemm_wreg.o conv288.o /tmp/ffn_abl_$v.o outer.o outer_pairs.o tied.o ops.o se3.o favor.o -o ../librfmi_ffnabl$v.so
done
