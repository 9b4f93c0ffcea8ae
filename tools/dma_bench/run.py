"""L2 -> LDS throughput of one CU with global_load_lds (tools/dma_bench/dma_bench.hip): GB/s per CU against waves issuing, pieces in
flight per wave and piece shape; plain global loads into registers beside it.  Run on the GPU box after build.sh."""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libdmabench.so"))
lib.dma_bench.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
src = torch.randint(0, 255, (2 << 20,), device="cuda", dtype=torch.uint8)   # 2 MB: L2-resident
sink = torch.zeros(4, device="cuda", dtype=torch.int32)
def run(grid, nw, pieces, depth, mode, stride=768):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.dma_bench(C.c_void_p(src.data_ptr()), src.numel(), grid, nw, pieces, depth, mode, stride, C.c_void_p(sink.data_ptr()), st)
    assert f() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 3 * 1e-3
    return nw * pieces * 1024 / t / 1e9      # GB/s per CU (one workgroup per CU)
names = {0: "1 KB contiguous -> LDS", 1: "16 rows x 64 B -> LDS", 2: "1 KB contiguous -> registers"}
for grid in (256, 32):
    print(f"== {grid} workgroups (one per CU)")
    for mode in (0, 1, 2):
        for nw in (1, 2, 4, 8):
            row = []
            for depth in (2, 4, 8, 16):
                if mode != 2 and nw * depth * 1024 > 150 * 1024: row.append("   -  "); continue
                row.append(f"{run(grid, nw, 20000, depth, mode):6.1f}")
            print(f"{names[mode]:30s} waves {nw}: GB/s per CU at 2/4/8/16 pieces in flight per wave: " + " ".join(row), flush=True)
