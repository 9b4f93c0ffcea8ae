for d in 0 1 2 4 8 3 6 9 11 15; do echo "RF_GEMM_DBG=$d"; RF_GEMM_DBG=$d ONLY="pair qkv,msa FF1,msa FF2,pair attn out" CFGS=0 timeout -k 10 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu; done
