#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh <tag>`: turns gpurun_out/prof_<tag>_* into profiles/<tag>_*_kernel_stats.{md,csv}, profiles/<tag>_pmc_report.md
# and profiles/traffic.json (run here, in the repo root).
TAG=${1:-r04}
S="rocprofv3 --kernel-trace --stats --output-format csv --"
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --no-train-leg --no-e2e-leg"
R=${TAG#r}; R="Round $((10#$R))"
python tools/stats_md.py gpurun_out/prof_${TAG}_step ${TAG}_step "$R, denoising step (f16 headline mode, B = 64 CFG step)" "$S $B" 4 "denoising step" ddim_step 5 | tail -1
python tools/stats_md.py gpurun_out/prof_${TAG}_parity ${TAG}_parity "$R, denoising step in parity mode (f16 x 3)" "$S $B --precision parity" 4 "denoising step" ddim_step 5 | tail -1
python tools/stats_md.py gpurun_out/prof_${TAG}_train ${TAG}_train "$R, training step (BASELINE config 2: B = 64, bf16; forward + L1 + backward + AdamW/EMA)" "$S python3 tools/bench_train.py" 5 "training step" adamw_ema_kernel | tail -1
python tools/stats_md.py gpurun_out/prof_${TAG}_svit ${TAG}_svit "$R, set-ViT style encoder (4 x 512^2 style images per sample, B = 64, bf16)" "$S python3 tools/bench_svit.py bf16 64" | tail -1
python tools/stats_md.py gpurun_out/prof_${TAG}_svit_fp8 ${TAG}_svit_fp8 "$R, set-ViT style encoder with MX-fp8 attention operands (B = 64)" "$S python3 tools/bench_svit.py fp8 64" | tail -1
python tools/stats_md.py gpurun_out/prof_${TAG}_swin ${TAG}_swin "$R, Swin-V2-T embedder (128 images of 512^2, bf16)" "$S python3 tools/bench_swin.py bf16 128 32" | tail -1
python tools/pmc_report.py gpurun_out/prof_${TAG}_pmc_mfma gpurun_out/prof_${TAG}_pmc_fetch gpurun_out/prof_${TAG}_pmc_write ${TAG} | tail -3
python tools/pmc_attention.py gpurun_out/prof_${TAG}_pmc_lsa gpurun_out/prof_${TAG}_pmc_lsa_fp8 ${TAG} | tail -2
