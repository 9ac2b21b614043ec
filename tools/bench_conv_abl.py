import os, sys, subprocess
shapes = "UP 1024|L2 1024->1024 @8 B128|L1 512->512 @16 B128|L0 256"
for dbg, name in [(0, "full"), (256, "all blocks on one (cache-hot) tile"), (320, "hot tile, no epilogue"), (2, "no patch DMA"), (1, "no weight DMA")]:
    env = dict(os.environ, STEDM_CONV_DBG=str(dbg), BENCH_FILTER=shapes, BENCH_DMA="1")
    out = subprocess.run([sys.executable, "tools/bench_conv.py", "bf16"], env=env, capture_output=True, text=True).stdout
    print(f"--- dbg={dbg}: {name}")
    for l in out.splitlines():
        if "us" in l and "SUM" not in l: print("   ", l)
