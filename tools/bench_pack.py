"""A/B timing of the fragment-order weight packs across library builds: python tools/bench_pack.py libA.so libB.so ..."""
import ctypes as C, sys, time
import torch
dev = torch.device("cuda:0")
P, I, L = C.c_void_p, C.c_int, C.c_long
cases = [("fwd 1024x1024x3x3", 1024, 1024, 3), ("fwd 512x1536x3x3", 512, 1536, 3), ("fwd 128x128x3x3", 128, 128, 3), ("fwd 3072x1024x1x1", 3072, 1024, 1)]
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    lib.stedm_pack_conv_weight_frag.argtypes = [P, P, I, I, I, I, P]
    has_s = hasattr(lib, "stedm_pack_conv_weight_strided")
    if has_s:
        lib.stedm_pack_conv_weight_strided.argtypes = [P, L, L, I, P, P, P, I, I, I, I, P]
    print(path)
    for name, co, ci, ks in cases:
        w = torch.randn(co, ci, ks, ks, device=dev)
        out = torch.empty(((co + 127) // 128, ci // 16, ks * ks, 4, 64, 8), dtype=torch.int16, device=dev)
        def run(): lib.stedm_pack_conv_weight_frag(w.data_ptr(), out.data_ptr(), co, ci, ks, 1, None)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"  {name:22s} {dt * 1e6:8.1f} us  {w.numel() * 6 / dt / 1e9:7.1f} GB/s")
        if has_s and ks == 3:
            taps = 9
            out2 = torch.empty(((ci + 127) // 128, co // 16, taps, 4, 64, 8), dtype=torch.int16, device=dev)
            def run2(): lib.stedm_pack_conv_weight_strided(w.data_ptr(), taps, ci * taps, 1, None, None, out2.data_ptr(), ci, co, ks, 1, None)
            for _ in range(3): run2()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): run2()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print(f"  dgrad {name[4:]:16s} {dt * 1e6:8.1f} us  {w.numel() * 6 / dt / 1e9:7.1f} GB/s")
    if has_s:
        dy = torch.randn(4096, 1024, device=dev)
        out3 = torch.empty((8, 4096 // 16, 1, 4, 64, 8), dtype=torch.int16, device=dev)
        def run3(): lib.stedm_pack_conv_weight_strided(dy.data_ptr(), 1, 1024, 0, None, None, out3.data_ptr(), 1024, 4096, 1, 1, None)
        for _ in range(3): run3()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run3()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"  dyT 4096x1024          {dt * 1e6:8.1f} us  {dy.numel() * 6 / dt / 1e9:7.1f} GB/s")
