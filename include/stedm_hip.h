/*
 * stedm_hip.h — C ABI of libstedm_hip.so, the MI355X (gfx950) implementation of STEDM's
 * denoising hot path.
 *
 * The reference (OettlM/STEDM) has no FFI: its hot path is a sequence of stock PyTorch ops
 * behind Python classes. Each entry point below therefore replaces a *PyTorch op
 * sequence* of the reference, cited as file:line under /root/reference/. The host side
 * (the stedm_amd python package) mirrors the reference's module surface (UNetModel, DDIMSampler, sViT, Agg_*)
 * and binds these symbols with ctypes; see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; stedm_last_error() gives the text
 *     (thread-local). Nothing is allocated for the caller; all buffers are caller-owned DEVICE
 *     pointers unless a parameter is documented as host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). All launches are
 *     asynchronous on that stream and are hipGraph-capturable (no allocation / sync inside).
 *   - activations inside the U-Net are NHWC fp32: x[b][y][x][c]. The two I/O convs translate
 *     from/to the reference's NCHW at the boundary.
 *   - "mm dtype" selects the MFMA operand format: STEDM_F16 / STEDM_BF16; "npass" 1 = single
 *     product, 3 = split-precision (hi*hi + hi*lo + lo*hi) used for fp32-parity.
 */
#ifndef STEDM_HIP_H
#define STEDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEDM_ABI_VERSION 12

#define STEDM_F16 0
#define STEDM_BF16 1

/* conv geometry modes */
#define STEDM_CONV_S1 0   /* 3x3 (or 1x1) stride 1, pad ks/2                                  */
#define STEDM_CONV_DOWN 1 /* 3x3 stride 2 pad 1 : Downsample.op       openaimodel.py:156-173  */
#define STEDM_CONV_UP 2   /* nearest x2 then 3x3 pad 1 : Upsample      openaimodel.py:122-132  */
#define STEDM_CONV_UP_SUBPIXEL 3 /* same operator evaluated as 4 output-parity 2x2 convs on the low-res input with
                                  * pre-summed taps (weights from stedm_pack_conv_weight_up): 4/9 of the MACs; DMA path */
#define STEDM_CONV_S2D 4  /* Downsample.op (3x3 stride 2 pad 1) on space-to-depth planes: src16 = [B][H/2][W/2][4*cin] from
                           * stedm_space_to_depth16 (channel block py*2+px holds pixel (2y+py, 2x+px)), evaluated as a stride-1
                           * conv with 2x2 taps (offsets -1, 0) over 4*cin channels; w_frag from stedm_pack_conv_weight_s2d_frag
                           * (7 of the 16 (tap, parity) blocks are zero). Register-streamed kernel only (single product): Hin/Win/c1
                           * describe the planes (H/2, W/2, 4*cin), ks = 3, w_hi may be NULL. */

int stedm_abi_version(void);
const char* stedm_last_error(void);
/* number of CUs of the current device (host query; used for launch heuristics / tests) */
int stedm_device_cus(void);

/* ---- fp16 operand range guard ------------------------------------------------------------------
 * The reference computes in fp32 (train_diff.py:48; convert_module_to_f16 is a no-op, openaimodel.py:25-29). In the `f16` and `parity`
 * (fp16 hi + lo) modes the MFMA operand planes are fp16: a residual-stream value beyond 65 504 would become inf in the un-normalised planes
 * (the operand of ResBlock.skip_connection openaimodel.py:247-254 / 288, of Upsample.conv :122-132 and of Downsample.op :156-173) and NaN one
 * layer later. Every kernel that rounds such values to fp16 ORs a site bit into `flag_words[0]` (device memory, 4 x u32, owned by the caller,
 * must outlive every launch and every captured graph) when a value it wrote has all exponent bits set (inf or NaN). bf16 planes are never
 * flagged (fp32 exponent range). The host reads and clears the word at its own synchronisation points (end of a sampling loop, after a
 * forward when asked) and raises. NULL switches the guard off. Set per device (the current one). */
#define STEDM_F16G_RAW 1        /* un-normalised planes written by stedm_gn_apply16c (skip_connection operand)          */
#define STEDM_F16G_NORM 2       /* normalised planes (GroupNorm output; bounded by sqrt(n) |gamma| + |beta|)             */
#define STEDM_F16G_CONV_OUT16 4 /* 16-bit side output of a convolution epilogue (Upsample operand, qkv planes)           */
#define STEDM_F16G_S2D 8        /* stedm_space_to_depth16 (Downsample operand)                                           */
#define STEDM_F16G_CAST 16      /* stedm_gn_apply16 / stedm_gn_chan_stats16 plain conversions                            */
#define STEDM_F16G_CONV_SRC 32  /* fp32-source convolution loader (hi/lo split in the kernel)                            */
int stedm_f16_guard_set(void* flag_words);

/* ---- weight packing (one-time, at load) -------------------------------------------------- */
/* OIHW fp32 conv weight [cout][cin][ks][ks] -> MFMA operand planes [cout][ks*ks][cin] (16-bit).
 * w_lo may be NULL (single-pass only). Replaces nothing in the reference (layout change only). */
int stedm_pack_conv_weight(const float* w_oihw, void* w_hi, void* w_lo, int cout, int cin, int ks,
                           int mm_dtype, void* stream);
/* Upsample conv weights for STEDM_CONV_UP_SUBPIXEL: OIHW 3x3 fp32 -> [4 parities (py*2+px)][cout][4 taps (a*2+b)][cin]
 * with W_eff[py][px][a][b] = sum of the 3x3 taps that read the same low-res pixel (rows: py=0 -> {0},{1,2}; py=1 -> {0,1},{2}). */
int stedm_pack_conv_weight_up(const float* w_oihw, void* w_hi, void* w_lo, int cout, int cin, int mm_dtype,
                              void* stream);
/* OIHW 3x3 or 1x1 fp32 -> MFMA-fragment order [ceil(cout/128)][cin/16][ks*ks taps][4][64 lanes][8] 16-bit (single product):
 * the layout the register-streamed kernels read with one coalesced 16-B load per lane and fragment. Passed to
 * stedm_conv_igemm as w_frag (optional; the kernel falls back to w_hi through LDS when it is NULL). */
int stedm_pack_conv_weight_frag(const float* w_oihw, void* out, int cout, int cin, int ks, int mm_dtype, void* stream);
/* The same for STEDM_CONV_UP_SUBPIXEL: OIHW 3x3 fp32 -> [4 parities][ceil(cout/128)][cin/16][4 taps][4][64][8] with the
 * pre-summed taps of stedm_pack_conv_weight_up (single product; cin %% 32 == 0). */
int stedm_pack_conv_weight_up_frag(const float* w_oihw, void* out, int cout, int cin, int mm_dtype, void* stream);
/* Operands of STEDM_CONV_S2D. space_to_depth16: NHWC fp32 [B][H][W][C] (H, W even) -> 16-bit planes [B][H/2][W/2][4*C]
 * (hi, and lo = x - hi when out_lo != NULL). pack_conv_weight_s2d_frag: OIHW 3x3 fp32 -> fragment order
 * [ceil(cout/128)][4*cin/16][4 taps (a*2+b)][4][64][8] of the equivalent 2x2 conv: tap (a, b) of parity block (py, px) is
 * W[dy][dx] with dy = {(0,1):0, (1,0):1, (1,1):2}[(a, py)] (none for (0,0)), same for dx. cin %% 4 == 0. */
int stedm_space_to_depth16(const float* x, int C, int B, int H, int W, void* out_hi, void* out_lo, int mm_dtype, void* stream);
/* OIHW 3x3 (ks 3) or 1x1 (ks 1) fp32 -> fragment order of the 16x16x32 MFMA kind: [ceil(cout/128)][cin/32][ks*ks taps][8 column
 * fragments][64 lanes][8]; lane l of column fragment c holds W[n = 128 tn + 16 c + (l & 15)][tap][ci], g = l >> 4:
 * ks 3: ci = 32 chunk + 16 (g & 1) + 8 (g >> 1) + e (the k-group -> (plane, 16-B piece) assignment of conv_rs.inc RS_3X3M);
 * ks 1: ci = 32 chunk + 8 g + e (the fused skip phase reads 128-B rows in natural piece order).
 * Element (n, ci, tap) is read at w[n*sn + ci*sc + tap'] with tap' = flip ? taps - 1 - tap : tap (sn = cin*taps, sc = taps, flip = 0
 * for a plain OIHW filter). */
int stedm_pack_conv_weight_frag16(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int ks, int mm_dtype, void* stream);
/* The 3x3 form for the 3-product mode (npass = 3): out = [2][total] 16-bit elements, the fragment stream of hi = round(w) followed by the
 * stream of lo = round(w - hi), total = ceil(cout / 128) * (cin / 32) * 9 * 4096. Passed as stedm_conv_args.w_frag16 with npass = 3, the plain
 * stride-1 3x3 problems from 128 input channels run on the register-streamed kernel (three MFMAs per fragment pair); w_hi / w_lo must still be
 * given: every other problem takes the LDS-operand kernels as before. */
int stedm_pack_conv_weight_frag16_hl(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int mm_dtype, void* stream);
/* ... of a 1x1 filter (ABI 9; the 3-product modes' skip_connection openaimodel.py:254, qkv / proj_out :343-346 on the register-streamed kernel):
 * out = [2][ceil(cout / 128)][cin / 32][1][8][512] 16-bit (hi stream, then lo stream), passed as stedm_conv_args.w_frag16 with ks = 1. */
int stedm_pack_conv_weight_frag16_hl1(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int mm_dtype, void* stream);
/* ... of the Upsample's sub-pixel form (openaimodel.py:122-132; 4 output parities x 2x2 pre-summed taps, as stedm_pack_conv_weight_up_frag) and of
 * the Downsample's space-to-depth form (:156-173; as stedm_pack_conv_weight_s2d_frag), w OIHW 3x3 fp32 (ABI 9):
 * out = [2][4 parities][ceil(cout / 128)][cin / 32][4][8][512] resp. [2][ceil(cout / 128)][4 cin / 32][4][8][512] 16-bit (hi stream, then lo),
 * passed as stedm_conv_args.w_frag16 with mode STEDM_CONV_UP_SUBPIXEL resp. STEDM_CONV_S2D and npass = 3. */
int stedm_pack_conv_weight_up_frag16_hl(const float* w, void* out, int cout, int cin, int mm_dtype, void* stream);
int stedm_pack_conv_weight_s2d_frag16_hl(const float* w, void* out, int cout, int cin, int mm_dtype, int pad_br, void* stream);
int stedm_pack_conv_weight_s2d_frag(const float* w_oihw, void* out, int cout, int cin, int mm_dtype, int pad_br, void* stream);
/* [rows][cols] fp32 -> [cols][rows] fp32 (Linear weights are consumed K-major). */
int stedm_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);

/* ---- GroupNorm statistics -> per-(sample, channel) affine ---------------------------------- */
/* GroupNorm32 / normalization(): util.py:199-216 (32 groups, biased variance, fp32, eps 1e-5);
 * attention.py:76-77 uses eps 1e-6. Input is the *virtual channel concat* [x1 | x2] of two NHWC
 * tensors (x2 may be NULL, c2 = 0), which removes th.cat([h, hs.pop()], 1) openaimodel.py:800.
 * x2 is indexed with batch (b % x2_bmod) when x2_bmod > 0 (shared encoder skips under CFG).
 * Writes scale[b][c] = rstd*gamma[c], shift[b][c] = beta[c] - mean*rstd*gamma[c], c in [0,c1+c2). */
int stedm_gn_scale_shift(const float* x1, int c1, const float* x2, int c2, int x2_bmod,
                         const float* gamma, const float* beta, float eps, int groups, int B, int HW,
                         float* scale, float* shift, void* stream);

/* Two-kernel form used by the DMA convolution path. (1) stats: per-(sample, pixel slab, group) sum and sum of
 * squares of the virtual concat [x1 | x2] (coalesced full-row reads) written to stats[B][nslab][groups][2] with
 * nslab = stedm_gn_nslab(c1+c2, HW); no atomics: apply adds the slabs in order, results are bitwise reproducible. (2) apply: y = act(GroupNorm(x)) written ONCE as 16-bit NHWC operand planes [B][HW][c1+c2]
 * (hi, and lo = y - hi when out_lo != NULL) — the concat is materialised in 16-bit, the conv then streams it by
 * LDS-DMA with no per-tile re-normalisation. act: 0 none, 1 SiLU. gamma == NULL: plain conversion (no norm). */
int stedm_gn_nslab(int C, int HW);
/* Producer-side statistics (used by UNetModel): chan_stats[B][nslab][C][2] fp32 = per-(sample, 256-pixel slab, channel)
 * {sum, sum of squares}, nslab = stedm_gn_chan_nslab(HW) = ceil(HW/256). Written by the convolution epilogue
 * (stedm_conv_args.chan_stats) or by stedm_gn_chan_stats for tensors of other producers. stedm_gn_apply16c folds them into
 * the group statistics of the virtual concat [x1 | x2] (x2 / cs2 indexed with b % x2_bmod), in a fixed order, and writes
 * act(GroupNorm(x)) as 16-bit planes; raw_hi/raw_lo (optional) receive the plain 16-bit conversion of [x1 | x2] from the
 * same read (operand of the ResBlock's 1x1 skip_connection, openaimodel.py:254). */
int stedm_gn_chan_nslab(int HW);
/* nslab: slot count of chan_stats; 0 or stedm_gn_chan_nslab(HW): runs of 256 pixels, otherwise nslab runs of ceil(HW/nslab). */
int stedm_gn_chan_stats(const float* x, int C, int B, int HW, int nslab, float* chan_stats, void* stream);
/* The same pass with the plain 16-bit conversion of x on the way (out_hi, and out_lo = x - hi when != NULL; NHWC [B][HW][C]): a gradient
 * tensor of the training backward needs both — the channel sums are the bias gradient (what autograd's sum over (N, H, W) of
 * F.conv2d's grad_output gives the reference, openaimodel.py:288), the planes the operand of its dgrad / wgrad. */
int stedm_gn_chan_stats16(const float* x, int C, int B, int HW, int nslab, float* chan_stats, void* out_hi, void* out_lo, int mm_dtype,
                          void* stream);
/* nslab1 / nslab2: slot counts of cs1 / cs2. ANY partition of a sample's pixels into slots serves (the consumer adds all slots):
 * 256-pixel runs (3x3 / 1x1 epilogues), (tile, output parity) pairs (sub-pixel upsample), row pairs (stedm_conv_in). */
int stedm_gn_apply16c(const float* x1, int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                      const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                      void* out_hi, void* out_lo, void* raw_hi, void* raw_lo, int mm_dtype, void* stream);
/* The same pass, also leaving mean_rstd [B][groups][2] = {mean, rstd} of every (sample, group) when != NULL: the statistics the pass folds
 * anyway, kept for the training backward (what autograd saves of F.group_norm, util.py:214-216) instead of a second fold (stedm_gn_fold). */
int stedm_gn_apply16c_mr(const float* x1, int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                         const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                         void* out_hi, void* out_lo, void* raw_hi, void* raw_lo, float* mean_rstd, int mm_dtype, void* stream);
/* The same pass when the h half of the concat already sits in the raw plane as 16-bit values, written there by the producing convolution
 * (stedm_conv_args.out16_hi + out16_stride; no fp32 tensor of it exists): raw_hi [B][HW][c1 + c2] holds x1 in channels [0, c1) on entry
 * and receives the plain conversion of x2 (fp32, [B or x2_bmod][HW][c2]) in [c1, c1 + c2); out_hi receives act(GroupNorm([x1|x2])).
 * 4 bytes per element of the h half instead of 8 (fp32 read + two 16-bit writes). cs1: the statistics the producer's epilogue left.
 * Single-product modes; c1 and c1 + c2 multiples of 8. Replaces the same reference lines as stedm_gn_apply16c. */
int stedm_gn_apply16c_x16(int c1, const float* cs1, int nslab1, const float* x2, int c2, const float* cs2, int nslab2, int x2_bmod,
                          const float* gamma, const float* beta, float eps, int groups, int act, int B, int HW,
                          void* out_hi, void* raw_hi, int mm_dtype, void* stream);
int stedm_gn_stats(const float* x1, int c1, const float* x2, int c2, int x2_bmod, int groups, int B, int HW,
                   double* stats, void* stream);
int stedm_gn_apply16(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* gamma,
                     const float* beta, float eps, int groups, int act, const double* stats, int B, int HW,
                     void* out_hi, void* out_lo, int mm_dtype, void* stream);

/* ---- fused implicit-GEMM convolution on MFMA ----------------------------------------------- */
typedef struct stedm_conv_args {
  const float* src1; /* NHWC [B][Hin][Win][c1]                                                 */
  const float* src2; /* NHWC [B or x2_bmod][Hin][Win][c2] or NULL : concat-free second source  */
  int32_t c1, c2, src2_bmod;
  int32_t B, Hin, Win;
  int32_t mode;      /* STEDM_CONV_*                                                           */
  int32_t ks;        /* 1 or 3                                                                 */
  const float* scale; /* [B][c1+c2] or NULL : GroupNorm apply fused into the A-tile load       */
  const float* shift;
  int32_t act;       /* 0 none, 1 SiLU (after the affine)                                      */
  const void* w_hi;  /* packed [cout][ks*ks][c1+c2] 16-bit                                     */
  const void* w_lo;  /* low plane (npass == 3) or NULL                                         */
  const float* bias; /* [cout] or NULL                                                         */
  const float* emb;  /* [*][cout] or NULL : h += emb[b*emb_bstride + n]  openaimodel.py:277-286 */
  int32_t emb_bstride;
  const float* res;  /* NHWC [B][Hout][Wout][cout] or NULL : residual add  openaimodel.py:288   */
  float* out;        /* NHWC [B][Hout][Wout][cout]                                             */
  int32_t cout;
  int32_t npass;     /* 1 or 3                                                                 */
  int32_t mm_dtype;  /* STEDM_F16 / STEDM_BF16                                                 */
  /* DMA path: when src16_hi != NULL the A operand is read from these pre-normalised 16-bit NHWC planes
   * [B][Hin][Win][c1+c2] (from stedm_gn_apply16) and src1/src2/scale/shift/act are ignored.              */
  const void* src16_hi;
  const void* src16_lo;
  /* epilogue extras (DMA path): act_out 0 none, 2 exact (erf) GELU applied after bias/emb/res; when out16_hi != NULL
   * the result is ALSO/ONLY written as 16-bit planes [M][cout] (hi, lo = v - hi); `out` may then be NULL.        */
  int32_t act_out;
  void* out16_hi;
  void* out16_lo;
  const void* w_frag; /* optional: fragment-order weights (stedm_pack_conv_weight_frag) for 3x3, npass 1, DMA path */
  float* chan_stats;  /* optional: [B][chan_nslab][cout][2] per-(sample, slot, channel) sum and sum of squares of `out` (the next
                       * GroupNorm's statistics, see stedm_gn_apply16c); needs out != NULL. chan_nslab = 0 means
                       * stedm_gn_chan_nslab(Hout*Wout) slots of 256 pixels; STEDM_CONV_UP_SUBPIXEL fills 4 * ceil(Hin*Win/256) slots
                       * (tile x output parity) from its epilogue when chan_nslab says so; any other count is filled by an extra pass */
  /* Fused skip_connection (optional, 3x3 stride 1, single product, needs w_frag): out = conv3x3(src16) + conv1x1(src16b) +
   * bias + bias_b — the ResBlock tail `skip_connection(x) + h` (openaimodel.py:254, 288) in one kernel. src16b_hi: raw 16-bit
   * planes [B][Hin][Win][cb] of the block input (raw output of stedm_gn_apply16c), cb %% 64 == 0; w_frag_b: the 1x1 weights in
   * fragment order (stedm_pack_conv_weight_frag, ks 1); res must be NULL. stedm_conv_fused_skip_ok() tells whether the
   * fused kernel covers a given problem; stedm_conv_igemm fails when asked for a fusion it cannot run. */
  const void* src16b_hi;
  const void* w_frag_b;
  const float* bias_b;
  int32_t cb;
  /* Optional workspace (fp32): lets a 3x3 convolution whose grid would leave CUs idle split its K range over k = 2, 4, 8 or 16
   * blocks per tile (the largest k with ws_floats >= k * B*Hout*Wout*cout that is needed); the partial tiles are summed in a fixed
   * order by a reduce kernel (bitwise reproducible). */
  float* ws;
  int64_t ws_floats;
  int32_t chan_nslab; /* slot count of chan_stats (see there) */
  const void* w_frag16; /* optional: the 3x3 weights in the fragment order of v_mfma_f32_16x16x32 (stedm_pack_conv_weight_frag16; cin %% 32 == 0;
                         * npass = 3: the hi + lo streams of stedm_pack_conv_weight_frag16_hl).
                         * When given, the plain 3x3 stride-1 single-product problems of the register-streamed kernel run on that MFMA
                         * shape (same tiles, same LDS image; the chip holds a higher clock on it on real data) */
  const void* w_frag_b16; /* with w_frag16 and the fused skip_connection: the 1x1 weights from stedm_pack_conv_weight_frag16(ks = 1) */
  int32_t pad_br;     /* STEDM_CONV_S2D only. 0: the stride-2 conv pads 1 on every side (openaimodel.py:164-166). 1: it pads bottom and
                       * right only — F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0), the VQ encoder's Downsample (model.py:59-76):
                       * the 2x2 taps of the space-to-depth form then sit at offsets (0, +1); weights from
                       * stedm_pack_conv_weight_s2d_frag(..., pad_br = 1) */
  /* Optional: the GroupNorm (+ SiLU) that consumes `out` — ResBlock.out_layers[0:2] after in_layers' convolution, openaimodel.py:236-241,
   * 275-287. gn_out16 != NULL asks for gn_act(GroupNorm(out; gn_gamma, gn_beta, gn_eps, gn_groups)) as 16-bit planes [B][Hout][Wout][cout]
   * (the operand planes of the next convolution, what stedm_gn_apply16c would write) besides `out`. Needs out, chan_stats, stride 1, and
   * either a single-product mode or gn_out16_lo; gn_out16 must not be the planes the convolution reads (src16_*): a tile's epilogue may write it while others still load. The split-K reduce pass writes them itself when one of its workgroups owns whole groups of a sample
   * (Hout * Wout <= 256, 32 %% (cout / gn_groups) == 0); otherwise the call ends with the stedm_gn_apply16c pass. */
  const float* gn_gamma;
  const float* gn_beta;
  float gn_eps;
  int32_t gn_groups;
  int32_t gn_act;    /* 0 none, 1 SiLU */
  void* gn_out16;
  float* gn_mr;      /* optional: [B][gn_groups][2] = {mean, rstd} of that GroupNorm (kept by a training forward for the backward) */
  int32_t gn_only;   /* 1: nothing but that GroupNorm reads `out` (inference: h of ResBlock._forward) — a launch whose epilogue writes
                      * gn_out16 itself may then leave `out` unwritten (4 of the 6 bytes per element of that store walk); `out` must
                      * still be a valid buffer, the other forms fill it. chan_stats is always written. */
  void* gn_out16_lo; /* npass == 3 (round 4): the lo planes of that GroupNorm output (value - hi, rounded): with it the 3-product modes get the
                      * same epilogue / reduce-pass GroupNorm as the single-product ones. NULL otherwise. */
  /* Optional (round 4, ABI 9): the LSA attention's operand planes straight from the to_qkv GEMM's epilogue (vit_set.py:52-57: to_qkv, chunk(3),
   * 'b n (h d) -> b h n d', q * temperature.exp()). qkv_q != NULL: the launch is a 1x1 GEMM over M = nb * qkv_T rows (B = 1, Hin = 1,
   * Win = M) with cout = 3 * qkv_heads * 64; column block s of heads * 64 is q (s = 0), k (1), v (2). Instead of `out` / `out16_*` (both NULL)
   * the epilogue writes q * qkv_qscale -> qkv_q [nb * heads][qkv_Tp][64], k -> qkv_k (same shape), v TRANSPOSED -> qkv_vt
   * [nb * heads][64][qkv_Tp], all 16-bit of mm_dtype; rows / columns t >= qkv_T are never written (the caller zeroed them once). Needs a
   * single-product mode, fragment-order weights (the register-streamed kernel is the only one with this epilogue: stedm_conv_rs_ok tells),
   * qkv_T and qkv_Tp even, heads even. Replaces the [M][3 * heads * 64] 16-bit output + stedm_qkv_pack16. */
  void* qkv_q;
  void* qkv_k;
  void* qkv_vt;
  int32_t qkv_T, qkv_Tp, qkv_heads;
  float qkv_qscale;
  /* Optional (round 4, ABI 9): a LayerNorm over the output ROW in the epilogue of a 1x1 GEMM whose rows fit one N-tile (cout <= 128) — Swin-V2's
   * res-post-norm `x = x + norm(proj(attn))` / `x + norm(mlp)` at 96 channels and the patch embedding's norm. ln_gamma != NULL:
   * out = LayerNorm(conv + bias; ln_gamma, ln_beta, ln_eps) (+ ln_res, fp32 [M][cout], may alias `out`), written as fp32 `out` and / or 16-bit
   * `out16_hi`. Needs a single-product mode, no res / emb / chan_stats / act_out / out16_lo, cout %% 4 == 0; only the register-streamed kernel has
   * this epilogue (stedm_conv_rs_ok tells). Replaces the GEMM's fp32 output + stedm_swin_ln. */
  const float* ln_gamma;
  const float* ln_beta;
  const float* ln_res;
  float ln_eps;
  /* Optional (round 5, ABI 10): elements per pixel row of the out16_hi plane (0: cout). With it a convolution writes its 16-bit output into
   * channels [0, cout) of a WIDER plane - the raw plane [B][Hout][Wout][cout + c_skip] of the decoder's th.cat([h, hs.pop()])
   * (openaimodel.py:800) that the next ResBlock's skip_connection reads and stedm_gn_apply16c_x16 normalises - and `out` may be NULL together
   * with chan_stats != NULL (the statistics are those of the fp32 values before their rounding): in inference nothing but that GroupNorm and
   * that 1x1 read the tensor, so no fp32 copy of it is ever stored. Register-streamed kernel, single-product modes, no out16_lo
   * (stedm_conv_rs_ok tells; split-K launches take it too). */
  int32_t out16_stride;
  /* Optional (round 5, ABI 12): the GroupNorm of gn_* in the epilogue also where a sample spans 2 .. 4 tiles of 256 pixels (the 32 x 32 level at
   * 128 channels: ResBlock.out_layers[0:2] behind in_layers' convolution, openaimodel.py:236-241). The tiles of a sample exchange their channel
   * sums inside the launch: gn_coop [B][4][128][2] 8-byte words {tag, fp32 bits} that every tile stores write-through and its partners poll
   * until the tag equals *gn_coop_epoch (a device word the caller advances once before every forward — never 0, frozen in no graph), then
   * every tile normalises its own staged rows. gn_coop must be zero before its first use and belongs to ONE call site (a second launch of the
   * same forward needs its own words); a spin that gives up ORs a code into *gn_coop_tmo (the caller checks it where it checks the fp16
   * guard) and the planes of that launch are garbage. Taken when cout == 128 (one N-tile), no split K, single product; the call ends with
   * the stedm_gn_apply16c pass as before otherwise. */
  void* gn_coop;
  const uint32_t* gn_coop_epoch;
  uint32_t* gn_coop_tmo;
} stedm_conv_args;
/* Replaces: GN->SiLU->conv3x3(+bias)(+emb)(+skip) of ResBlock._forward openaimodel.py:268-288,
 * Downsample/Upsample convs (:122-132,:156-173), 1x1 skip_connection (:254), and the 1x1
 * Conv1d qkv / proj_out of AttentionBlock (:326,:334,:343-346). */
int stedm_conv_igemm(const stedm_conv_args* args, void* stream);
/* 1 when stedm_conv_igemm would run `args` (with src16b_hi / w_frag_b / cb set) as one fused kernel, else 0. No launch. */
int stedm_conv_fused_skip_ok(const stedm_conv_args* args);
/* 1 when the tiled 3x3 kernels of stedm_conv_igemm have a tiling for an Hout x Wout output grid (runs of 128 / 256 pixels aligned with the
 * image rows: power-of-two widths, in practice), else 0: the host then runs the convolution as stedm_im2col_rows16 + the 1x1 kind. */
int stedm_conv3x3_tiles_ok(int Hout, int Wout);
/* Row-major im2col of 16-bit NHWC planes for the generic-shape form of a 3x3 convolution (any H, W: what UNetModel.forward of the reference
 * accepts, openaimodel.py:761-806): dst [B][Ho][Wo][9 C], column tap * C + c = src [b][sy][sx][c] or 0 outside. mode 0: stride 1 pad 1
 * (Ho = Hs); 1: stride 2 pad 1 (Downsample.op :164-166, Ho = (Hs - 1) / 2 + 1); 2: nearest x2 then stride 1 pad 1 (Upsample :129-131,
 * Ho = 2 Hs). The GEMM over K = 9 C then reads the ordinary [cout][tap][cin] planes of stedm_pack_conv_weight. C %% 8 == 0. */
int stedm_im2col_rows16(const void* src16, void* dst16, int B, int Hs, int Ws, int C, int mode, void* stream);
/* 1 when the register-streamed kernel (w_frag) takes this problem: w_hi / w_lo are then never read, so the caller may skip packing
 * them and pass any non-NULL w_hi. Same decision path as stedm_conv_igemm; nothing is launched. */
int stedm_conv_rs_ok(const stedm_conv_args* args);

/* ---- boundary convs (NCHW <-> NHWC) ------------------------------------------------------- */
/* input_blocks.0: conv3x3(cat([x, c_concat],1)) — DiffusionWrapper hybrid ddpm.py:1414-1417 +
 * openaimodel.py:542. x1 NCHW [B][c1][H][W], x2 NCHW [B or bmod][c2][H][W] (may be NULL);
 * w OIHW fp32 [cout][c1+c2][3][3]; out NHWC [B][H][W][cout]. Exact fp32 FMA. */
int stedm_conv_in(const float* x1, int c1, const float* x2, int c2, int x2_bmod, const float* w_oihw,
                  const float* bias, float* out, int B, int H, int W, int cout, float* chan_stats, void* stream);
/* chan_stats (optional, fast path only: c1+c2 <= 8, 256 %% cout == 0, H even): [B][H/2][cout][2] channel partials of `out`, one slot
 * per pair of image rows; returns 3 without writing when the fast path does not apply (the caller then uses stedm_gn_chan_stats). */
/* out: GN->SiLU->conv3x3 to out_channels openaimodel.py:729-733, 806. src NHWC [B][H][W][c] (c %% 16 == 0); GroupNorm statistics
 * from the producer-side channel partials chan_stats[B][nslab][c][2] (stedm_conv_args.chan_stats / stedm_gn_chan_stats);
 * w HWIO fp32 [3][3][c][cp], cp = 4 (cout <= 4) or 8, zero-padded (the OIHW weight permuted once by the host); c %% 32 == 0;
 * out NCHW. Exact fp32 FMA. */
int stedm_conv_out(const float* src, int c, const float* chan_stats, int nslab, const float* gamma, const float* beta,
                   float eps, int groups, const float* w_hwio, const float* bias, float* out, int B, int H, int W,
                   int cout, void* stream);

/* ---- embedding path ---------------------------------------------------------------------- */
/* timestep_embedding util.py:151-171 + time_embed openaimodel.py:529-534,774-775.
 * t int64 [B]; freqs fp32 [mc/2] (host-built table, uploaded once); w0t [mc][ted], w2t [ted][ted]
 * are TRANSPOSED Linear weights; emb [B][ted]; ws: scratch of B*(mc+ted) floats. */
int stedm_time_embed(const int64_t* t, const float* freqs, const float* w0t, const float* b0,
                     const float* w2t, const float* b2, float* emb, float* ws, int B, int mc, int ted,
                     void* stream);
/* emb_layers = SiLU -> Linear of every ResBlock at once (openaimodel.py:231-237,277):
 * out[b][n] = bias[n] + sum_k silu(emb[b][k]) * wt[k][n]; wt is [k][ntot] (layers concatenated). */
int stedm_emb_proj(const float* emb, const float* wt, const float* bias, float* out, int B, int k, int ntot,
                   void* stream);

/* out[b][n] = act_out(bias[n] + sum_k act_in(x[b][k]) * wt[k][n]); act: 0 none, 1 SiLU, 2 ReLU. wt K-major [k][n].
 * Used for Agg_Linear's MLP (agg_blocks.py:14-18, 28-31). */
int stedm_linear(const float* x, const float* wt, const float* bias, float* out, int B, int k, int n, int act_in,
                 int act_out, void* stream);

/* ---- middle attention (QKVAttentionLegacy) -------------------------------------------------- */
/* openaimodel.py:378-394. qkv [B][T][heads*3*ch] with channel = h*3*ch + {q:0,k:ch,v:2ch} + c;
 * out [B][T][heads*ch]; scale ch^-1/4 on q and k, softmax in fp32. */
int stedm_attn_legacy(const float* qkv, float* out, int B, int T, int heads, int ch, void* stream);
/* The same for T == 64 tokens and ch in {32, 64, 128} on MFMA (single-product modes): q, k, v and the softmax weights are rounded
 * to the 16-bit operand type, logits / softmax / normalisation stay fp32; writes the 16-bit operand plane [B][64][heads*ch] that
 * proj_out's 1x1 reads (no fp32 intermediate). qkv_is16: qkv is itself the 16-bit plane written by the qkv convolution's epilogue
 * (stedm_conv_args.out16_hi) instead of fp32 - the same rounding, half the bytes.
 * Also T = 64 n <= 4096 tokens with ch in {64, 128} and qkv_is16 = 1 (the middle block at 64x64 / 128x128 latents): 64-key tiles with
 * an online softmax around the same products. Any other shape returns an error (the caller uses stedm_attn_legacy). */
int stedm_attn_legacy16(const void* qkv, int qkv_is16, void* out16, int B, int T, int heads, int ch, int mm_dtype, void* stream);

/* ---- DDIM update with rescaled classifier-free guidance ----------------------------------- */
/* ddim.py:179-184 (CFG + std rescale over dims (C,H), unbiased) and :195-210 (x0 / dir / noise).
 * x, e_c, e_u, noise, x_prev, pred_x0: NCHW fp32 [B][C][H][W]. e_u NULL => no guidance.
 * coefs: DEVICE table [nsteps][4] = {a_t, a_prev, sigma_t, sqrt(1-a_t)}; step_idx: DEVICE int
 * (NULL => row 0) so that a captured graph can be replayed for every step. noise may be NULL
 * (sigma*noise == 0). pred_x0 may be NULL. x_prev may alias x. */
int stedm_ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise,
                    const float* coefs, const int32_t* step_idx, float cfg_scale, float rescale_phi,
                    float* x_prev, float* pred_x0, int B, int C, int H, int W, void* stream);
/* *step_idx += delta (device-side loop counter for graph replay). */
int stedm_step_advance(int32_t* step_idx, int delta, void* stream);
/* t_buf[0..B) = ts_table[*step_idx] : ts = torch.full((b,), step) of ddim.py:141, device-side so that one
 * captured graph serves every step. ts_table: DEVICE int64 [nsteps] (ddim_timesteps, ascending). */
int stedm_step_set_t(const int64_t* ts_table, const int32_t* step_idx, int64_t* t_buf, int B, void* stream);

/* ---- style path: set-ViT encoder (networks/vit_set.py), aggregation blocks, layout rescaler -------------------- */
/* SPT vit_set.py:84-107 + token assembly :175-186. img [B][ns][H][W][3] fp32 -> x [B][ntok+2][dim]:
 * x[:,2+t] = Linear(LayerNorm(patch_t)) + pos[2+t]; x[:,0] = cls + pos[0]; x[:,1] = pos[1] (zero time token).
 * wt is the TRANSPOSED Linear weight [patch_dim][dim]; patch feature index = (p1*p + p2)*(3*ns) + c*ns + s. */
int stedm_svit_patch_embed(const float* img, int B, int ns, int H, int W, int patch, const float* ln_w,
                           const float* ln_b, float eps, const float* wt, const float* bias, const float* pos,
                           const float* cls, float* x, int dim, void* stream);
/* The same embedding in three steps, so that its Linear runs on MFMA: (1) patch gather + LayerNorm -> 16-bit operand planes
 * [B*ntok][patch_dim]; (2) stedm_conv_igemm (1x1) with the Linear's weight and bias -> tok [B*ntok][dim] fp32; (3) tok_place:
 * x[:,2+t] = tok[t] + pos[2+t], x[:,0] = cls + pos[0], x[:,1] = pos[1]. */
int stedm_svit_patch_ln16(const float* img, int B, int ns, int H, int W, int patch, const float* ln_w, const float* ln_b,
                          float eps, void* out_hi, void* out_lo, int mm_dtype, void* stream);
int stedm_svit_tok_place(const float* tok, const float* pos, const float* cls, float* x, int B, int ntok, int dim, void* stream);
/* PreNorm LayerNorm vit_set.py:14-20 -> 16-bit operand planes [rows][dim] for the MFMA GEMMs (out_lo may be NULL). */
int stedm_ln_apply16(const float* x, const float* gamma, const float* beta, float eps, void* out_hi, void* out_lo,
                     long rows, int dim, int mm_dtype, void* stream);
/* to_qkv output [B][T][3*heads*64] ('(h d)' per chunk, vit_set.py:53-54) -> q/k [B*heads][Tp][64] (q scaled by
 * qscale = exp(temperature) * log2(e): stedm_lsa_flash works in the log2 domain, vit_set.py:56; rows >= T zero) and
 * V^T [B*heads][64][Tp]; Tp multiple of 128. qkv: fp32 rows, or (qkv_is16, single-product modes) the 16-bit plane of type mm_dtype that the
 * to_qkv GEMM wrote as its out16. */
int stedm_qkv_pack(const void* qkv, int qkv_is16, float qscale, void* q_hi, void* q_lo, void* k_hi, void* k_lo, void* vt_hi,
                   void* vt_lo, int B, int T, int Tp, int heads, int mm_dtype, void* stream);
/* LSA attention vit_set.py:56-66: softmax over keys of q.k with the DIAGONAL masked to -FLT_MAX, times v; flash-style
 * on MFMA (head dim 64). q.k must be log2(e) times the reference's logits (see stedm_qkv_pack): p = exp2(s - max).
 * out planes [B][T][heads*64] ('b h n d -> b n (h d)'). */
int stedm_lsa_flash(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                    const void* vt_lo, void* out_hi, void* out_lo, int B, int T, int Tp, int heads, int npass,
                    int mm_dtype, void* stream);
/* MX-fp8 variant of the LSA attention (BASELINE config 5 "fp8 MFMA attention"; same algorithm as stedm_lsa_flash) on the block-scaled matrix
 * instruction v_mfma_scale_f32_32x32x64_f8f6f4: OCP e4m3 bytes with one E8M0 power-of-two scale (byte = exponent + 127) per 32 elements of
 * the contraction. stedm_qkv_pack_mx8: qkv [M][3*heads*64] (fp32, or 16-bit values of type mm_dtype when qkv_is16 — the qkv GEMM's out16) ->
 *   q8, k8 [B*heads][Tp][64] bytes (q times qscale; rows >= T zero), qs, ks [B*heads][Tp][2]: scale of (token, channel half);
 *   vt8 [B*heads][64][Tp]: V^T with the keys of every 64-key tile permuted — position 32 s + 16 h + m holds key 32 s + 8 (m >> 2) + 4 h + (m & 3),
 *   the contraction order of the kernel's P fragment — and vs [B*heads][Tp/32][64]: scale of (32-key sub-tile, channel).
 * stedm_lsa_flash_mx8 runs the flash attention on them (P as e4m3 of 16 p with the scale 2^-4; fp32 logits / sums / accumulation) and
 * writes the 16-bit plane [B][T][heads*64] (type mm_dtype) that to_out consumes. One pack pass, no tensor-wide amax. A precision experiment:
 * its deviation from the reference is reported, not asserted at 1e-3. Tp % 128 == 0. */
int stedm_qkv_pack_mx8(const void* qkv, int qkv_is16, float qscale, void* q8, void* qs, void* k8, void* ks, void* vt8, void* vs, int B, int T,
                       int Tp, int heads, int mm_dtype, void* stream);
int stedm_lsa_flash_mx8(const void* q8, const void* qs, const void* k8, const void* ks, const void* vt8, const void* vs, void* out16, int B,
                        int T, int Tp, int heads, int mm_dtype, void* stream);
/* Train-mode dropout of the style ViT — the reference runs S_ZSS_DM.get_input, and with it the agg block, inside the training step with the
 * LightningModule in train mode (networks/s_zss_dm.py:45-60), so nn.Dropout is live at networks/vit_set.py:187 (after pos_embedding,
 * emb_dropout), :43/:62 (attention probabilities), :49 (after to_out's Linear) and :28-30 (after the FeedForward's GELU and after its second
 * Linear); conf/style_agg/svit.yaml: 0.1 each. torch's generator stream cannot be reproduced, so the masks are counter-based and specified
 * here (oracle/dropmask.py restates them): keep iff u16 >= thr16 = lrint(p * 65536), kept values scaled by 1 / (1 - p), p in [0, 1).
 *   elementwise sites (stedm_dropout_rows): element e of the tensor's linear index takes the 16-bit field (e & 7) — half (j & 1) of output
 *     word j >> 1 — of Philox4x32-10(counter = (lo32(e >> 3), hi32(e >> 3), site, 0), key = (lo32 seed, hi32 seed)).
 *     out[e] = drop(src[e]) (+ res[e]); written as fp32 (out) and / or 16-bit operand planes (out_hi, out_lo; type mm_dtype).
 *     out may alias src or res.
 *   attention site (stedm_lsa_flash_drop = stedm_lsa_flash with dropout on the normalised probabilities): one xorshift128 stream (Marsaglia
 *     2003: t = x ^ x << 11; x, y, z = y, z, w; w ^= w >> 19 ^ t ^ t >> 8) per (query q, sample-head bh, key half h), state (x, y, z, w) =
 *     Philox4x32-10(counter = (q, bh, site, h), key = seed); for every 64-key tile kt = 0, 1, ... the stream yields 16 words, the first the
 *     most significant bit-plane of 32 16-bit uniforms; uniform i = 16 sub + e belongs to key 64 kt + 32 sub + (e & 3) + 8 (e >> 2) + 4 h.
 * `site` separates the sites of one forward (stedm_amd/style.py: 8 * layer + kind); `seed` is drawn per forward from torch's generator. */
int stedm_dropout_rows(const float* src, const float* res, float* out, void* out_hi, void* out_lo, long n, float p,
                       unsigned long long seed, unsigned site, int mm_dtype, void* stream);
int stedm_lsa_flash_drop(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                         const void* vt_lo, void* out_hi, void* out_lo, int B, int T, int Tp, int heads, int npass,
                         int mm_dtype, float p, unsigned long long seed, unsigned site, void* stream);
/* pool (0 mean, 1 cls, 2 sum) over tokens (+ c_old) -> mlp_head LayerNorm + Linear, vit_set.py:191-206. wt [dim][ncls].
 * ws (optional, ws_floats >= B * 2 * dim): lets the token pooling run as slab partials over ~1024 blocks before the per-sample head
 * (a batch of 8 alone would read its 34 MB of tokens on 8 of the 256 CUs); fixed summation order either way. */
int stedm_svit_head(const float* x, int B, int T, int dim, int pool, const float* c_old, const float* ln_w,
                    const float* ln_b, float eps, const float* wt, const float* bias, float* out, int ncls, float* ws, long ws_floats,
                    void* stream);
/* GEGLU attention.py:37-44: g [M][2*I] fp32 (value | gate) -> value * gelu_erf(gate) as 16-bit planes [M][I]. */
int stedm_geglu16(const float* g, void* out_hi, void* out_lo, long M, int I, int mm_dtype, void* stream);
/* Agg_Mean (mode 0) / Agg_Max (mode 1) over the set: feats [B*n][F] -> out [B][F]; agg_blocks.py:52,73. */
int stedm_agg_reduce(const float* feats, float* out, int B, int n, int F, int mode, void* stream);
/* SpatialRescaler encoders/modules.py:123-130: n_stages x bilinear 1/2 (== box mean for divisible sizes) then bias-free
 * 1x1 conv w [cout][cin] (NULL: none). x NCHW [B][cin][H][W] -> out NCHW [B][cout][H>>n][W>>n]. */
int stedm_spatial_rescale(const float* x, const float* w, float* out, int B, int cin, int cout, int H, int W,
                          int n_stages, void* stream);

/* ---- Swin-Transformer-V2 style embedder ---------------------------------------------------------
 * Replaces the non-GEMM pieces of torchvision's swin_v2_t (torchvision==0.18.1, third party; call sites networks/s_zss_dm.py:19-20,
 * networks/agg_blocks.py:28,49,70); every Linear / the patch Conv2d runs through stedm_conv_igemm (1x1). Parity unpinned.
 * features[0][0] Conv2d(3, 96, 4, 4) as a GEMM: img [N][3][H][W] with element strides (sn, sc, sh, sw) -> 16-bit operand rows
 * [N*(H/4)*(W/4)][64], column k = c*16 + ky*4 + kx (the flattened OIHW weight), columns 48..63 zero. */
int stedm_swin_patch16(const float* img, long sn, long sc, long sh, long sw, int N, int H, int W, void* out_hi, void* out_lo,
                       int mm_dtype, void* stream);
/* out = res + LayerNorm(y) over rows of `dim` (SwinTransformerBlockV2's post-norm residual x + norm(f(x)); res NULL: plain LayerNorm);
 * fp32 rows (out, may be NULL) and / or 16-bit operand planes (out_hi / out_lo, may be NULL) of row stride ld16 >= dim elements (columns
 * dim..ld16-1 are not written: a caller that pads K to the GEMM kernel's 64-channel chunks zeroes them once). res may alias out. dim <= 768. */
int stedm_swin_ln(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                  void* out_lo, long rows, int dim, int ld16, int mm_dtype, void* stream);
/* stedm_swin_ln with train-mode stochastic depth (torchvision.ops.StochasticDepth(p, "row") on both residual branches of
 * SwinTransformerBlockV2; swin_v2_t: p rises linearly to 0.2 over the 12 blocks): out = res + gate[row / rows_per_gate] * LayerNorm(y),
 * gate [rows / rows_per_gate] = bernoulli(1 - p) / (1 - p) per image, drawn by the caller (stedm_amd/swin.py). The reference runs the
 * embedder inside the training step in train mode (networks/s_zss_dm.py:45-60, networks/agg_blocks.py:28,49,70). */
int stedm_swin_ln_gated(const float* y, const float* gamma, const float* beta, float eps, const float* res, float* out, void* out_hi,
                        void* out_lo, long rows, int dim, int ld16, const float* gate, int rows_per_gate, int mm_dtype, void* stream);
/* torchvision shifted_window_attention with ShiftedWindowAttentionV2's cosine logits, 8 x 8 windows, head dim 32, on MFMA (npass 1: single
 * product; 3: hi/lo split products, the parity mode): qkv [N*H*W][3C] fp32 in token order (bias included, k bias zeroed) ->
 * softmax(normalize(q) normalize(k)^T * scale[h] + rpb + mask) v as the 16-bit plane(s) [N*H*W][C] `proj` consumes. Cyclic shift, window
 * partition, F.pad rows (q = bias_q, k = 0, v = bias_v) and their inverses are index arithmetic; normalisation, logits and softmax are fp32.
 * bias_kzero [3C]; scale [heads] = exp(min(logit_scale, log 100)); rpb [heads][query][key] = 16 sigmoid(cpb_mlp(relative_coords_table))
 * [relative_position_index]. A side no larger than the window is not shifted. qkv16 (single-product modes; then qkv is NULL): the same rows as
 * 16-bit values of type mm_dtype — the qkv GEMM's out16 — which halves the largest stream of stage 1. */
int stedm_swin_window_attn(const float* qkv, const void* qkv16, const float* bias_kzero, const float* scale, const float* rpb, void* out_hi, void* out_lo,
                           int ld16, int N, int H, int W, int C, int heads, int shift, int npass, int mm_dtype, void* stream);
/* PatchMergingV2's input: x [N][H][W][C] fp32 -> 16-bit operand rows [N*ceil(H/2)*ceil(W/2)][4C] = [x(0,0) | x(1,0) | x(0,1) | x(1,1)]
 * (zero beyond an odd side). */
int stedm_swin_merge16(const float* x, int N, int H, int W, int C, void* out_hi, void* out_lo, int mm_dtype, void* stream);
/* ShiftedWindowAttentionV2.get_relative_position_bias: cpb [ntab][heads] = cpb_mlp(relative_coords_table) (two stedm_linear calls),
 * index [64*64] int64 = relative_position_index (query-major) -> rpb [heads][query][key] = 16 sigmoid(cpb[index]). */
int stedm_swin_rpb(const float* cpb, const long* index, float* rpb, int heads, int ntab, void* stream);
/* AdaptiveAvgPool2d(1) over the tokens: x [N][T][C] -> out [N][C]. */
int stedm_swin_token_mean(const float* x, float* out, int N, int T, int C, void* stream);

/* Several fragment-order packs (stedm_pack_conv_weight_frag / _frag16 / the fragment part of _strided) in ONE launch: the training step
 * re-packs every convolution's weights after each optimizer step (forward order + flipped / transposed dgrad order, ~130 tensors), and the
 * individual packs were launch-bound. descs: DEVICE array of nd records
 *   { const float* w; void* out; long sn, sc; int cout, cin, taps (9 | 1), flip, m16 (0: 32x32x16 order, 1: 16x16x32 order), blk0; }   (56 bytes)
 * blk0 = first block of the tensor, ascending; its block count is ceil(cout/128) * (cin/16) * 2 (m16: ceil(cout/128) * (cin/32) * 4);
 * total_blocks = their sum. Outputs are bit-identical to the single-tensor entry points. */
int stedm_pack_frag_multi(const void* descs, int nd, int total_blocks, int mm_dtype, void* stream);

/* ---- training step: backward of the U-Net, loss, optimizer -------------------------------------
 * Replaces torch.autograd over UNetModel.forward (openaimodel.py:761-806) inside LatentDiffusion.p_losses (ddpm.py:1015-1048),
 * torch.optim.AdamW (ldm_diffusion.py:224-234) and LitEma.forward (ema.py:25-44). The convolution contractions of the backward
 * (dgrad, wgrad) run on stedm_conv_igemm itself: dgrad with the flipped/transposed filter, wgrad as the GEMM
 * dW[(tap,ci)][co] = sum_p col[(tap,ci)][p] * dYt[co][p] over the planes stedm_im2col_t16 writes. */
/* The packs of stedm_pack_conv_weight / _frag from a strided source: element (n, ci, tap) = w[n*sn + ci*sc + (flip ? taps-1-tap : tap)]
 * (dgrad filter = flipped + transposed OIHW parameter; dY^T of the wgrad GEMM from the NHWC gradient). NULL outputs are skipped. */
int stedm_pack_conv_weight_strided(const float* w, long sn, long sc, int flip, void* w_hi, void* w_lo, void* w_frag, int cout,
                                   int cin, int ks, int mm_dtype, void* stream);
/* chan partials (stedm_gn_chan_stats / conv epilogues) of the virtual concat [x1|x2] -> mean_rstd [B][groups][2]. */
int stedm_gn_fold(const float* cs1, int nslab1, int c1, const float* cs2, int nslab2, int c2, int groups, int B, int HW,
                  float eps, float* mean_rstd, void* stream);
/* Backward of act(GroupNorm32([x1|x2])) (util.py:199-216; act 1 = SiLU): dA [B][HW][C] -> dx (+ add, the block's residual
 * branch) into dx1 [B][HW][c1] / dx2 [B][HW][c2] (accN: accumulate), optional 16-bit planes of dx, dgamma/dbeta (acc_param).
 * ws: stedm_gn_bwd_ws_floats(B, HW, C, groups) floats. */
int stedm_gn_bwd(const float* x1, int c1, const float* x2, int c2, const float* mean_rstd, const float* gamma,
                 const float* beta, int groups, int act, const float* dA, const float* add, int B, int HW, float* ws,
                 float* dx1, int acc1, float* dx2, int acc2, void* dx16_hi, void* dx16_lo, int mm_dtype, float* dgamma,
                 float* dbeta, int acc_param, void* stream);
long stedm_gn_bwd_ws_floats(int B, int HW, int C, int groups);
/* 16-bit NHWC planes [B][Hs][Ws][C] -> transposed im2col [(tap*C + c)][Ppad] over the conv's output grid (mode 0 stride 1,
 * 1 nearest-2x upsample + conv (openaimodel.py:129-131), 2 stride 2 pad 1 (:164-166)); pixels >= P are zero. */
int stedm_im2col_t16(const void* src16, void* dst16, int B, int Hs, int Ws, int C, int ks, int mode, long Ppad, void* stream);
/* GEMM result dw [nsplit][taps][cin_ld][cout_ld] (nsplit split-K partials, summed in order) -> OIHW gradient [cout][cin][taps]
 * (accumulate: +=). */
int stedm_wgrad_to_oihw(const float* dw, float* grad, int cout, int cin, int taps, int cin_ld, int cout_ld, int accumulate,
                        int nsplit, void* stream);
/* Direct weight gradient of a stride-1 3x3 pad-1 convolution from the NHWC 16-bit planes (no im2col): x16 [B][H][W][Cin], dy16
 * [B][H][W][Cout] (bf16) -> part [ksplit][9][Cin][Cout] fp32 partials (then stedm_wgrad_to_oihw with nsplit = ksplit).
 * stedm_wgrad3x3_plan returns 1 when the shape is supported (W in {8,16,32,64}, H %% (64/W) == 0, Cin %% 128 == 0, Cout %% 64 == 0) and
 * the split it will use. */
int stedm_wgrad3x3_plan(int B, int H, int W, int Cin, int Cout, int* ksplit);
int stedm_wgrad3x3(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype,
                   void* stream);
/* stedm_wgrad3x3 with its split-K partials in the parameter's own order: part = [ksplit][Cout][Cin][3][3] fp32, 16-byte aligned. With
 * ksplit == 1 (stedm_wgrad3x3_plan) `part` may be the gradient tensor itself; otherwise stedm_sum_planes adds the slices (fixed order).
 * Replaces the weight gradient autograd computes for nn.Conv2d(3x3, stride 1) in ResBlock / Upsample (openaimodel.py:122-132, 214-254). */
int stedm_wgrad3x3_oihw(const void* x16, const void* dy16, float* part, int B, int H, int W, int Cin, int Cout, int mm_dtype, void* stream);
/* out[i] (+)= sum_z part[z * n + i] for z < nsplit, fixed order (n % 4 == 0, 16-byte aligned pointers) */
int stedm_sum_planes(const float* part, float* out, long n, int nsplit, int accumulate, void* stream);
/* The same for a 1x1 convolution (skip_connection, attention qkv / proj_out): x16 [P][Cin], dy16 [P][Cout] (bf16, P = B*H*W pixels) ->
 * part [ksplit][Cin][Cout] fp32 partials (then stedm_wgrad_to_oihw with taps = 1, nsplit = ksplit). stedm_wgrad1x1_plan returns 1 when the
 * shape is supported (P %% 64 == 0, Cin %% 128 == 0, Cout %% 128 == 0) and the split it will use. */
int stedm_wgrad1x1_plan(long P, int Cin, int Cout, int* ksplit);
int stedm_wgrad1x1(const void* x16, const void* dy16, float* part, long P, int Cin, int Cout, int mm_dtype, void* stream);
/* chan partials cs [B][nslab][C][2] -> per-sample channel sums per_sample[b*ld + c] (NULL: skip) and their batch total
 * total[c] (bias gradients; the per-sample sums are the gradient of the emb_layers output, openaimodel.py:277-280). */
int stedm_chan_sum_fold(const float* cs, int B, int nslab, int C, float* per_sample, long ld, float* total, int accumulate,
                        void* stream);
/* ... with the total also written to total2 (NULL: skip): two parameters that receive the same gradient — conv1.bias and
 * emb_layers[1].bias both add onto h (openaimodel.py:276-280), conv2.bias and skip_connection.bias onto the block output (:288). */
int stedm_chan_sum_fold2(const float* cs, int B, int nslab, int C, float* per_sample, long ld, float* total, int accumulate, float* total2,
                         void* stream);
/* backward of F.interpolate(scale 2, nearest): out [B][H][W][C] (+)= 2x2 block sums of in [B][2H][2W][C]. */
int stedm_sum2x2(const float* in, float* out, int B, int H, int W, int C, int accumulate, void* stream);
/* in [B][Ho][Wo][C] fp32 -> 16-bit planes [B][2Ho][2Wo][C], value at even positions, zero elsewhere (stride-2 dgrad). */
int stedm_zero_insert16(const float* in, void* hi, void* lo, int B, int Ho, int Wo, int C, int mm_dtype, void* stream);
/* SpatialTransformer backward pieces (ldm/modules/attention.py:196-261; the Linears and the attention go through the convolution / attention
 * backward entry points above). LayerNorm backward over rows of dim <= 2048 (BasicTransformerBlock.norm1/2/3): dx = add + rstd (dxh - mean(dxh)
 * - xh mean(dxh xh)) with dxh = dy gamma (add may be NULL; dx may alias add), dgamma / dbeta = column sums of dy xh / dy (accumulate: added onto
 * the tensors). ws: 2 * dim * stedm_ln_bwd_blocks(rows) floats. GEGLU backward (attention.py:37-44, exact erf GELU): g [M][2 I] (value | gate),
 * dh [M][I] -> dg [M][2 I]. */
int stedm_ln_bwd_blocks(long rows);
int stedm_ln_bwd(const float* x, const float* dy, const float* gamma, float eps, const float* add, float* dx, float* dgamma, float* dbeta,
                 float* ws, long rows, int dim, int accumulate, void* stream);
int stedm_geglu_bwd(const float* g, const float* dh, float* dg, long M, int I, void* stream);
/* backward of QKVAttentionLegacy (openaimodel.py:378-394): qkv, d_qkv [B][T][heads*3*ch]; d_out [B][T][heads*ch]. ws: workspace of
 * stedm_attn_legacy_bwd_ws_floats(B, T, heads) floats (0 = none: both T x T matrices stay in LDS, T <= 128). */
long stedm_attn_legacy_bwd_ws_floats(int B, int T, int heads);
int stedm_attn_legacy_bwd(const float* qkv, const float* d_out, float* d_qkv, int B, int T, int heads, int ch, float* ws, void* stream);
/* C = alpha op(A) op(B) + beta C (fp32; the embedding Linears' backward: rows = batch). ws (optional, ws_floats floats): partial
 * sums of the split-K form taken when the output is small and K long (fixed-order reduce). */
int stedm_gemm_f32(const float* A, long lda, int trans_a, const float* B, long ldb, int trans_b, float* C, long ldc, int M,
                   int N, int K, float alpha, float beta, float* ws, long ws_floats, void* stream);
/* mode 0: out = silu(x); mode 1: out = dy * silu'(x). */
int stedm_silu(const float* x, const float* dy, float* out, long n, int mode, void* stream);
/* q_sample (ddpm.py:277-280): out[b] = sqrt_ac[t[b]] * x0[b] + sqrt_1mac[t[b]] * noise[b]; n elements per sample, t int64 [B]
 * (tables: the fp32 buffers sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod of ddpm.py:155-156). Bit-exact vs the fp32 reference. */
int stedm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                   float* out, int B, long n, void* stream);
/* Per-sample normal noise on the device: out [rows][n] fp32, row i ~ N(0, 1) depending only on (seed, stream, sample id of row i) - x_T
 * (ddim.py:122) and the per-step noise of eta > 0 (ddim.py:206) of a rank's shard in a data-parallel prediction run, identical for any
 * world size (the reference's batch-shaped draw from the global generator is not). sample_ids: device int64 [rows], or NULL for
 * first_id + row. Philox4x32-10, counter {element / 4, stream, 0x4E524D4C, 0}, key {seed (low 32 bits), sample id}; Box-Muller on word
 * pairs (the parity tests hold a numpy restatement of this definition). */
int stedm_philox_normal(float* out, int rows, int n, const long* sample_ids, int first_id, unsigned long long seed, unsigned stream, void* stream_);

/* loss = mean|target - pred| (ddpm.py:282-295 'l1' + :1030-1040), d_pred = grad_scale * sign(pred - target) / n (NULL: skip).
 * ws: 1024 doubles. */
int stedm_l1_loss(const float* pred, const float* target, long n, float grad_scale, float* d_pred, double* ws, float* loss,
                  void* stream);
/* SpatialRescaler.channel_mapper weight gradient (encoders/modules.py:123-130 under autograd, cond_stage_trainable): x NCHW
 * [B][cin][H][W], d_out [B][cout][H>>n][W>>n] (= the c_concat slice of the U-Net's input gradient) -> dw [cout][cin]. ws: B*cin*cout floats. */
int stedm_spatial_rescale_wgrad(const float* x, const float* d_out, float* ws, float* dw, int B, int cin, int cout, int H, int W,
                                int n_stages, int accumulate, void* stream);
/* y = alpha x + beta y over n floats (n %% 4 == 0): gradient accumulation across micro-batches (Trainer(accumulate_grad_batches),
 * train_diff.py). */
int stedm_axpby_f32(const float* x, float* y, long n, float alpha, float beta, void* stream);
/* AdamW + EMA over many tensors: table [ntensors] of {float* p, const float* g, float* m, float* v, float* ema|NULL, long n};
 * block i updates elements [chunk_off[i], chunk_off[i] + 4096) of tensor chunk_tensor[i]. step counts from 1. */
int stedm_adamw_ema(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float ema_decay, float grad_scale, void* stream);
/* The same update for convolution weights (OIHW fp32, cout %% rows == 0, cin %% ciw == 0, 1 or 9 taps) that also refreshes the 16-bit
 * fragment-order copies of stedm_pack_frag_multi in the same pass (torch.optim.AdamW.step of modules/ldm_diffusion.py:224-234 followed by the
 * weight packs of the next forward / backward). descs [ndesc] of
 *   { float* p; const float* g; float* m; float* v; float* ema|NULL; int cout, cin, taps, blk0, nout, pad;
 *     { void* out; int transposed, flip, m16, f16; } o[4]; }                                   (160 bytes)
 * blk0 = first block of the tensor (ascending; a tensor takes (cout / rows) * (cin / ciw) blocks, stedm_adamw_ema_pack_piece), total_blocks their sum. Output k is the
 * pack of element (n, c, tap) = W[n][c][tap] (transposed = 0: stedm_pack_conv_weight_frag / _frag16 of the OIHW filter) or
 * W[c][n][taps-1-tap] (transposed = 1, flip = 1: the dgrad filter), m16: the 16x16x32 fragment order, f16: fp16 instead of bf16. */
int stedm_adamw_ema_pack_piece(int* rows, int* ciw);   /* the piece a block owns: a tensor takes (cout / rows) * (cin / ciw) blocks */
int stedm_adamw_ema_pack(const void* descs, int ndesc, int total_blocks, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float ema_decay, float grad_scale, void* stream);
/* The two passes above for a CAPTURED training step (hipGraph replay; the step of modules/ldm_diffusion.py:224-234 / ddpm.py:345-371 is
 * shape-static): the step-dependent scalars come from the device. sched [n][4] = { 1 - beta1^s, sqrtf(1 - beta2^s), LitEma decay of step s,
 * learning rate of step s } for a window of steps (built by the host with the arithmetic of the two calls above), *sched_idx = the row of the
 * step being run (advanced inside the captured step by stedm_step_advance). */
int stedm_adamw_ema_sched(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float beta1, float beta2, float eps,
                          float weight_decay, const float* sched, const int* sched_idx, float grad_scale, void* stream);
int stedm_adamw_ema_pack_sched(const void* descs, int ndesc, int total_blocks, float beta1, float beta2, float eps, float weight_decay,
                               const float* sched, const int* sched_idx, float grad_scale, void* stream);
/* LitEma.forward alone (ldm/modules/ema.py:25-44; on_train_batch_end, ddpm.py:369-371, runs it once per micro-batch, also on the
 * micro-batches of an accumulation window that do not step the optimizer): ema -= (1 - ema_decay) * (ema - p) over the same table
 * (entries without a shadow are skipped; g / m / v are not read). */
int stedm_ema_update(const void* table, const int* chunk_tensor, const long* chunk_off, int nchunks, float ema_decay, void* stream);

/* ---- first stage (VQ-f4 autoencoder; ldm/models/autoencoder.py:264-282, ldm/modules/diffusionmodules/model.py:368-568) ---------
 * Its ResnetBlocks, Up/Downsample convs and GroupNorms (eps 1e-6) run on stedm_conv_igemm / stedm_gn_apply16c; these are the rest.
 * VectorQuantizer2.forward of taming-transformers (un-vendored dependency, autoencoder.py:6,39-41,277) on the eval path: z NCHW
 * [B][e_dim][HW], codebook [n_e][e_dim] -> idx[B*HW] = argmin_n (|z|^2 + |e_n|^2 - 2 z.e_n) (fp32, no FMA contraction, sums left to
 * right, first index on ties: an integer result) and zq NCHW = z + (e_idx - z) (the straight-through form's forward value). */
int stedm_vq_nearest(const float* z, const float* codebook, int n_e, int e_dim, int B, long HW, long long* idx, float* zq, void* stream);
/* quant_conv / post_quant_conv (autoencoder.py:42-43, 268, 280): 1x1 conv over a few channels (cin, cout <= 16), NCHW -> NCHW,
 * w [cout][cin], bias [cout] or NULL. */
int stedm_conv1x1_nchw(const float* x, const float* w, const float* bias, float* out, int B, int cin, int cout, long HW, void* stream);
/* AttnBlock's softmax (model.py:166-169: softmax(w_ * c^-0.5, dim=2)): rows of x [rows][ld_in] (first n columns) -> softmax(scale * x) as
 * 16-bit operand planes [rows][ld_out] (hi, lo = v - hi or NULL), columns n..ld_out-1 zero. */
int stedm_softmax_rows16(const float* x, long ld_in, float scale, void* out_hi, void* out_lo, long rows, int n, long ld_out, int mm_dtype,
                         void* stream);

/* ---- image epilogue of predict_step (integer work, bit-exact) -------------------------------------------------------
 * modules/ldm_diffusion.py:93-95: ((clip(x, -1, 1).permute(0,2,3,1) + 1) * 127.5).astype(uint8): x NCHW fp32 -> out NHWC uint8. */
int stedm_image_to_uint8(const float* x, unsigned char* out, int B, int C, int H, int W, void* stream);
/* LDM_Diffusion.prepare_batch, modules/ldm_diffusion.py:52-56: seg NCHW [B][K][H][W] -> NHWC [B][H][W][2] = {class 0, sum of classes
 * 1..K-1}. */
int stedm_seg_merge(const float* seg, float* out, int B, int K, int H, int W, void* stream);
/* modules/ldm_diffusion.py:98: torch.argmax(segmentation, dim=-1).astype(uint8): seg [N][ncls] fp32 -> out [N] (first maximum). */
int stedm_argmax_u8(const float* seg, unsigned char* out, long N, int ncls, void* stream);

/* ---- HIP graph capture helpers (plumbing for the sampling loop) ---------------------------- */
int stedm_graph_begin(void* stream);
int stedm_graph_end(void* stream, void** graph_exec_out);
int stedm_graph_launch(void* graph_exec, void* stream);
int stedm_graph_destroy(void* graph_exec);

/* ---- diagnostics (no reference counterpart) ----
 * With STEDM_CONV_DBG & 1024 the bf16 register-streamed 3x3 kernel records 5 phase stamps (100 MHz clock) per block:
 * entry, tables built, first patch landed, main loop done, stores retired. Copies 8 x nblocks u64 to host memory. */
int stedm_debug_conv_stamps(unsigned long long* host_out, int nblocks);

#ifdef __cplusplus
}
#endif
#endif /* STEDM_HIP_H */
