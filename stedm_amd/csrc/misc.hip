// Error plumbing, weight packing, transposes, graph helpers, embedding path, DDIM update.
#include <stdarg.h>

#include "common.hpp"
#include "dropmask.hpp"
#include "sgemm.hpp"

namespace stedm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static unsigned* g_f16_guard[64] = {nullptr};
unsigned* f16_guard_flag() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return g_f16_guard[dev];
}
}  // namespace stedm
using namespace stedm;

extern "C" int stedm_f16_guard_set(void* flag_words) {
  int dev = 0;
  STEDM_HIP_TRY(hipGetDevice(&dev));
  STEDM_CHECK_ARG(dev >= 0 && dev < 64, "f16_guard_set: device index %d out of range", dev);
  STEDM_CHECK_ARG(((uintptr_t)flag_words & 15) == 0, "f16_guard_set: the flag words must be 16-byte aligned");
  stedm::g_f16_guard[dev] = reinterpret_cast<unsigned*>(flag_words);
  return 0;
}

extern "C" int stedm_abi_version(void) { return STEDM_ABI_VERSION; }
extern "C" const char* stedm_last_error(void) { return g_err; }
extern "C" int stedm_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    set_error("no HIP device");
    return -1;
  }
  return p.multiProcessorCount;
}

// ------------------------------------------------------------------------------------------------
// Weight packing: OIHW fp32 -> [cout][tap][cin] 16-bit hi (+ lo = round(w - float(hi))).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ hi, T* __restrict__ lo, int cout,
                                        int cin, int taps, long sn, long sc, int flip) {
  const long total = (long)cout * taps * cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    const long r = i / cin;
    const int tap = (int)(r % taps);
    const int co = (int)(r / taps);
    const float v = w[(long)co * sn + (long)ci * sc + (flip ? taps - 1 - tap : tap)];
    const T h = (T)v;
    hi[i] = h;
    if (lo) lo[i] = (T)(v - (float)h);
  }
}

extern "C" int stedm_pack_conv_weight(const float* w, void* w_hi, void* w_lo, int cout, int cin, int ks, int mm_dtype,
                                      void* stream) {
  STEDM_CHECK_ARG(w && w_hi, "pack_conv_weight: null pointer");
  STEDM_CHECK_ARG(ks == 1 || ks == 3, "pack_conv_weight: ks must be 1 or 3 (got %d)", ks);
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight: bad mm_dtype %d", mm_dtype);
  const long total = (long)cout * cin * ks * ks;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (mm_dtype == STEDM_F16)
    pack_conv_weight_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)w_hi, (_Float16*)w_lo, cout, cin,
                                                                         ks * ks, (long)cin * ks * ks, ks * ks, 0);
  else
    pack_conv_weight_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)w_hi, (__bf16*)w_lo, cout, cin,
                                                                       ks * ks, (long)cin * ks * ks, ks * ks, 0);
  STEDM_LAUNCH_CHECK();
  return 0;
}

template <typename T>
__global__ void pack_conv_weight_up_kernel(const float* __restrict__ w, T* __restrict__ hi, T* __restrict__ lo, int cout, int cin) {
  const long total = (long)4 * cout * 4 * cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    long r = i / cin;
    const int tap = (int)(r % 4); r /= 4;
    const int co = (int)(r % cout);
    const int par = (int)(r / cout);
    const int py = par >> 1, px = par & 1, a = tap >> 1, b = tap & 1;
    // 3x3 taps that land on low-res neighbour (a, b) for output parity (py, px)
    const int dy0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), dy1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int dx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), dx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float v = 0.f;
    for (int dy = dy0; dy <= dy1; ++dy)
      for (int dx = dx0; dx <= dx1; ++dx) v += w[((long)co * cin + ci) * 9 + dy * 3 + dx];
    const T h = (T)v;
    hi[i] = h;
    if (lo) lo[i] = (T)(v - (float)h);
  }
}

extern "C" int stedm_pack_conv_weight_up(const float* w, void* w_hi, void* w_lo, int cout, int cin, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && w_hi, "pack_conv_weight_up: null pointer");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_up: bad mm_dtype %d", mm_dtype);
  const long total = (long)16 * cout * cin;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (mm_dtype == STEDM_F16)
    pack_conv_weight_up_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)w_hi, (_Float16*)w_lo, cout, cin);
  else
    pack_conv_weight_up_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)w_hi, (__bf16*)w_lo, cout, cin);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// Fragment-order packing for the register-streamed 3x3 kernel (conv_rs.inc):
//   out[tn][chunk][tap][q][lane][e] = W[n = tn*128 + (q>>1)*64 + (q&1)*32 + (lane&31)][ci = chunk*16 + (lane>>5)*8 + e][tap]
// (rows beyond cout are zero). One wave-wide 16-B load = one MFMA B fragment, 1 KiB contiguous.
// One block = one (128-row tile tn, 16-channel chunk, 64-row half): the 64 x 16 x taps source values go through LDS so that both the
// gather from the source (runs along its contiguous index: (ci, tap) for an OIHW filter, (n, tap) for the transposed forms) and the
// 16-B fragment stores are coalesced.
// M16: the fragment order of the 16x16x32 MFMA kind (stedm_pack_conv_weight_frag16): blocks of 32 rows x 32 channels,
//   out[tn][chunk32][tap][c][lane][e] = W[n = tn*128 + c*16 + (lane&15)][ci = chunk*32 + kofs(lane>>4) + e][tap], kofs(g) = 16 (g&1) + 8 (g>>1) for a
//   3x3 (plane, piece), 8 g for a 1x1.
constexpr int kPackTileFloats = 64 * (16 * 9 + 1);     // the largest tile of the four (taps, M16) forms: 64 rows x (16 x 9 + 1)

// block `bid` of one tensor's pack; `traw`: kPackTileFloats floats of LDS
template <typename T, int taps, bool M16>
__device__ __forceinline__ void pack_frag_block(const float* __restrict__ w, T* __restrict__ out, const int cout, const int cin, const long sn,
                                                const long sc, const int flip, const int bid, float* traw, T* __restrict__ out_lo = nullptr) {
  typedef T V8 __attribute__((ext_vector_type(8)));
  constexpr int NR = M16 ? 32 : 64, KC = M16 ? 32 : 16, NSUB = 128 / NR;     // rows and channels per block, blocks per 128-row tile
  constexpr int PITCH = KC * taps + 1;
  static_assert(NR * PITCH <= kPackTileFloats, "pack tile");
  float (*tile)[PITCH] = reinterpret_cast<float (*)[PITCH]>(traw);
  const int nch = cin / KC;
  const int half = bid % NSUB;
  const int chunk = (bid / NSUB) % nch, tn = (bid / NSUB) / nch;
  const int n0 = tn * 128 + half * NR, ci0 = chunk * KC;
  const int per = NR * KC * taps;
  const bool n_fast = sn <= sc;       // which source index is contiguous
  // 16-B loads along the contiguous source run when the layout allows it (rows inside the matrix, 16-B aligned runs): the pack is a
  // pure stream, its speed is the bytes in flight
  const bool vec = n0 + NR <= cout && (n_fast ? (sn == taps && (sc * 4) % 16 == 0) : (sc == taps && (sn * 4) % 16 == 0)) &&
                   ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((n_fast ? NR : KC) * taps) % 4 == 0;
  if (vec) {
    const int run = (n_fast ? NR : KC) * taps;          // contiguous floats per outer index (ci for n_fast, n otherwise)
    const int nouter = n_fast ? KC : NR;
    for (int v4 = threadIdx.x; v4 < nouter * (run / 4); v4 += 256) {
      const int o = v4 / (run / 4), r0 = (v4 - o * (run / 4)) * 4;
      const float* src = n_fast ? w + (long)n0 * sn + (long)(ci0 + o) * sc + r0 : w + (long)(n0 + o) * sn + (long)ci0 * sc + r0;
      const float4 f = *reinterpret_cast<const float4*>(src);
      const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + j, a = r / taps, ts = r - a * taps;       // a = n_local (n_fast) or ci_local; ts = source tap
        const int tap = flip ? taps - 1 - ts : ts;
        if (n_fast) tile[a][o * taps + tap] = fv[j];
        else tile[o][a * taps + tap] = fv[j];
      }
    }
  } else
  for (int idx = threadIdx.x; idx < per; idx += 256) {
    int row, cil, tap;
    if (n_fast) { cil = idx / (NR * taps); const int r = idx - cil * NR * taps; row = r / taps; tap = r - row * taps; }
    else { row = idx / (KC * taps); const int r = idx - row * KC * taps; cil = r / taps; tap = r - cil * taps; }
    const int n = n0 + row;
    tile[row][cil * taps + tap] = n < cout ? w[(long)n * sn + (long)(ci0 + cil) * sc + (flip ? taps - 1 - tap : tap)] : 0.f;
  }
  __syncthreads();
  // stores: (tap, fragment of the block, lane) -> one 16-B vector of 8 consecutive channels
  for (int v = threadIdx.x; v < taps * 2 * 64; v += 256) {
    const int lane = v & 63, ql = (v >> 6) & 1, tap = v >> 7;
    int row, c8;
    if (M16) { const int g = lane >> 4; row = ql * 16 + (lane & 15); c8 = taps == 9 ? (g & 1) * 16 + (g >> 1) * 8 : g * 8; }
    else { row = ql * 32 + (lane & 31); c8 = (lane >> 5) * 8; }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)tile[row][(c8 + e) * taps + tap];
    const int q = half * 2 + ql;
    const long at = ((((long)tn * nch + chunk) * taps + tap) * (M16 ? 8 : 4) + q) * 512 + lane * 8;
    *reinterpret_cast<V8*>(out + at) = o;
    if (out_lo) {     // 3-product mode: the second stream holds w - hi
      V8 l;
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = (T)(tile[row][(c8 + e) * taps + tap] - (float)o[e]);
      *reinterpret_cast<V8*>(out_lo + at) = l;
    }
  }
}

template <typename T, int taps, bool M16 = false>
__global__ void __launch_bounds__(256) pack_conv_weight_frag_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, long total,
                                                                    long sn, long sc, int flip, T* __restrict__ out_lo = nullptr) {
  __shared__ float traw[kPackTileFloats];
  pack_frag_block<T, taps, M16>(w, out, cout, cin, sn, sc, flip, blockIdx.x, traw, out_lo);
}

// Many tensors in ONE launch (the training step re-packs every convolution's weights after each optimizer step: ~130 packs of a few
// microseconds each were launch-bound). descs[i].blk0 = first block of tensor i (ascending); a block finds its tensor by binary search.
struct PackDesc {
  const float* w;
  void* out;
  long sn, sc;
  int cout, cin, taps, flip, m16, blk0;
};
static_assert(sizeof(PackDesc) == 56, "PackDesc layout is part of the ABI (stedm_pack_frag_multi)");

template <typename T>
__global__ void __launch_bounds__(256) pack_frag_multi_kernel(const PackDesc* __restrict__ descs, const int nd) {
  __shared__ float traw[kPackTileFloats];
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  const int bid = blockIdx.x - d.blk0;
  T* out = reinterpret_cast<T*>(d.out);
  if (d.m16) {
    if (d.taps == 9) pack_frag_block<T, 9, true>(d.w, out, d.cout, d.cin, d.sn, d.sc, d.flip, bid, traw);
    else pack_frag_block<T, 1, true>(d.w, out, d.cout, d.cin, d.sn, d.sc, d.flip, bid, traw);
  } else {
    if (d.taps == 9) pack_frag_block<T, 9, false>(d.w, out, d.cout, d.cin, d.sn, d.sc, d.flip, bid, traw);
    else pack_frag_block<T, 1, false>(d.w, out, d.cout, d.cin, d.sn, d.sc, d.flip, bid, traw);
  }
}

extern "C" int stedm_pack_frag_multi(const void* descs, int nd, int total_blocks, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(descs && nd > 0 && total_blocks > 0, "pack_frag_multi: bad args");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_frag_multi: bad mm_dtype %d", mm_dtype);
  if (mm_dtype == STEDM_F16) pack_frag_multi_kernel<_Float16><<<total_blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const PackDesc*>(descs), nd);
  else pack_frag_multi_kernel<__bf16><<<total_blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const PackDesc*>(descs), nd);
  STEDM_LAUNCH_CHECK();
  return 0;
}

static void launch_pack_frag(const float* w, void* out, int cout, int cin, int taps, long total, long sn, long sc, int flip, int mm_dtype, int grid,
                             hipStream_t st) {
  if (mm_dtype == STEDM_F16) {
    if (taps == 9) pack_conv_weight_frag_kernel<_Float16, 9><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip);
    else pack_conv_weight_frag_kernel<_Float16, 1><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip);
  } else {
    if (taps == 9) pack_conv_weight_frag_kernel<__bf16, 9><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip);
    else pack_conv_weight_frag_kernel<__bf16, 1><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip);
  }
}

extern "C" int stedm_pack_conv_weight_frag(const float* w, void* out, int cout, int cin, int ks, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 16 == 0 && (ks == 1 || ks == 3), "pack_conv_weight_frag: bad args (cin %% 16, ks 1 or 3)");
  const int taps = ks * ks;
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_frag: bad mm_dtype %d", mm_dtype);
  const long total = (long)((cout + 127) / 128) * (cin / 16) * taps * 4 * 64 * 8;
  const int grid = ((cout + 127) / 128) * (cin / 16) * 2;
  launch_pack_frag(w, out, cout, cin, taps, total, (long)cin * taps, taps, 0, mm_dtype, grid, as_stream(stream));
  STEDM_LAUNCH_CHECK();
  return 0;
}

// Same packs from an arbitrarily strided source: element (n, ci, tap) = w[n*sn + ci*sc + (flip ? taps-1-tap : tap)]. The backward pass
// packs the dgrad filter (n = forward cin, ci = forward cout, taps reversed) straight from the OIHW parameter, and dY^T of the
// wgrad GEMM (n = channel, ci = pixel) straight from the NHWC gradient. Any of w_hi / w_lo / w_frag may be NULL.
extern "C" int stedm_pack_conv_weight_strided(const float* w, long sn, long sc, int flip, void* w_hi, void* w_lo, void* w_frag, int cout, int cin, int ks,
                                              int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && (w_hi || w_frag) && (ks == 1 || ks == 3), "pack_conv_weight_strided: bad args");
  STEDM_CHECK_ARG(!w_frag || cin % 16 == 0, "pack_conv_weight_strided: fragment order needs cin %% 16 == 0");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_strided: bad mm_dtype %d", mm_dtype);
  const int taps = ks * ks;
  hipStream_t st = as_stream(stream);
  if (w_hi) {
    const long total = (long)cout * cin * taps;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (mm_dtype == STEDM_F16) pack_conv_weight_kernel<_Float16><<<grid, 256, 0, st>>>(w, (_Float16*)w_hi, (_Float16*)w_lo, cout, cin, taps, sn, sc, flip);
    else pack_conv_weight_kernel<__bf16><<<grid, 256, 0, st>>>(w, (__bf16*)w_hi, (__bf16*)w_lo, cout, cin, taps, sn, sc, flip);
  }
  if (w_frag) {
    const long total = (long)((cout + 127) / 128) * (cin / 16) * taps * 4 * 64 * 8;
    const int grid = ((cout + 127) / 128) * (cin / 16) * 2;
    launch_pack_frag(w, w_frag, cout, cin, taps, total, sn, sc, flip, mm_dtype, grid, st);
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

// Sub-pixel upsample weights (see pack_conv_weight_up_kernel) in fragment order: [4 parities][tn][cin/16][4 taps][4][64][8]
template <typename T>
__global__ void pack_conv_weight_up_frag_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, long per_parity) {
  const int nch = cin / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * per_parity; i += (long)gridDim.x * blockDim.x) {
    const int par = (int)(i / per_parity);
    const long ii = i - par * per_parity;
    const int e = (int)(ii & 7);
    const int lane = (int)((ii >> 3) & 63);
    const int q = (int)((ii >> 9) & 3);
    long r = ii >> 11;
    const int tap = (int)(r & 3); r >>= 2;
    const int chunk = (int)(r % nch);
    const int tn = (int)(r / nch);
    const int n = tn * 128 + (q >> 1) * 64 + (q & 1) * 32 + (lane & 31);
    const int ci = chunk * 16 + (lane >> 5) * 8 + e;
    const int py = par >> 1, px = par & 1, a = tap >> 1, b = tap & 1;
    const int dy0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), dy1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int dx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), dx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float v = 0.f;
    if (n < cout)
      for (int dy = dy0; dy <= dy1; ++dy)
        for (int dx = dx0; dx <= dx1; ++dx) v += w[((long)n * cin + ci) * 9 + dy * 3 + dx];
    out[i] = (T)v;
  }
}

extern "C" int stedm_pack_conv_weight_up_frag(const float* w, void* out, int cout, int cin, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 32 == 0, "pack_conv_weight_up_frag: bad args (cin %% 32)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_up_frag: bad mm_dtype %d", mm_dtype);
  const long per = (long)((cout + 127) / 128) * (cin / 16) * 4 * 4 * 64 * 8;
  const int grid = (int)((4 * per + 255) / 256 < 8192 ? (4 * per + 255) / 256 : 8192);
  if (mm_dtype == STEDM_F16) pack_conv_weight_up_frag_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)out, cout, cin, per);
  else pack_conv_weight_up_frag_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)out, cout, cin, per);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_pack_conv_weight_frag16(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int ks, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 32 == 0 && cout > 0 && (ks == 1 || ks == 3), "pack_conv_weight_frag16: bad args (cin %% 32, ks 1 or 3)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_frag16: bad mm_dtype %d", mm_dtype);
  const int taps = ks * ks;
  const long total = (long)((cout + 127) / 128) * (cin / 32) * taps * 8 * 64 * 8;
  const int grid = ((cout + 127) / 128) * (cin / 32) * 4;
  hipStream_t st = as_stream(stream);
  if (mm_dtype == STEDM_F16) {
    if (taps == 9) pack_conv_weight_frag_kernel<_Float16, 9, true><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip);
    else pack_conv_weight_frag_kernel<_Float16, 1, true><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip);
  } else {
    if (taps == 9) pack_conv_weight_frag_kernel<__bf16, 9, true><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip);
    else pack_conv_weight_frag_kernel<__bf16, 1, true><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip);
  }
  STEDM_LAUNCH_CHECK();
  return 0;
}

// hi and lo fragment streams of the 3-product mode: out = [2][total] elements, total = ceil(cout / 128) * (cin / 32) * 9 * 8 * 512
extern "C" int stedm_pack_conv_weight_frag16_hl(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 32 == 0 && cout > 0, "pack_conv_weight_frag16_hl: bad args (cin %% 32)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_frag16_hl: bad mm_dtype %d", mm_dtype);
  const long total = (long)((cout + 127) / 128) * (cin / 32) * 9 * 8 * 64 * 8;
  const int grid = ((cout + 127) / 128) * (cin / 32) * 4;
  hipStream_t st = as_stream(stream);
  if (mm_dtype == STEDM_F16)
    pack_conv_weight_frag_kernel<_Float16, 9, true><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip, (_Float16*)out + total);
  else
    pack_conv_weight_frag_kernel<__bf16, 9, true><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip, (__bf16*)out + total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ... of a 1x1 filter (skip_connection, qkv, proj_out in the 3-product modes): out = [2][ceil(cout / 128)][cin / 32][1][8][512]
extern "C" int stedm_pack_conv_weight_frag16_hl1(const float* w, long sn, long sc, int flip, void* out, int cout, int cin, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 32 == 0 && cout > 0, "pack_conv_weight_frag16_hl1: bad args (cin %% 32)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_frag16_hl1: bad mm_dtype %d", mm_dtype);
  const long total = (long)((cout + 127) / 128) * (cin / 32) * 8 * 64 * 8;
  const int grid = ((cout + 127) / 128) * (cin / 32) * 4;
  hipStream_t st = as_stream(stream);
  if (mm_dtype == STEDM_F16)
    pack_conv_weight_frag_kernel<_Float16, 1, true><<<grid, 256, 0, st>>>(w, (_Float16*)out, cout, cin, total, sn, sc, flip, (_Float16*)out + total);
  else
    pack_conv_weight_frag_kernel<__bf16, 1, true><<<grid, 256, 0, st>>>(w, (__bf16*)out, cout, cin, total, sn, sc, flip, (__bf16*)out + total);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// STEDM_CONV_S2D weights: the stride-2 3x3 as a 2x2 conv over the 4 parity blocks of the space-to-depth planes, fragment order
template <typename T>
__global__ void pack_conv_weight_s2d_frag_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, long total, int pad_br) {
  const int nch = 4 * cin / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const int q = (int)((i >> 9) & 3);
    long r = i >> 11;
    const int tap = (int)(r & 3); r >>= 2;
    const int chunk = (int)(r % nch);
    const int tn = (int)(r / nch);
    const int n = tn * 128 + (q >> 1) * 64 + (q & 1) * 32 + (lane & 31);
    const int cc = chunk * 16 + (lane >> 5) * 8 + e;          // channel of the planes: parity block * cin + c
    const int par = cc / cin, c = cc - par * cin;
    const int py = par >> 1, px = par & 1, a = tap >> 1, b = tap & 1;
    // symmetric pad 1: plane rows (y-1, y) hold image rows 2y-2 .. 2y+1, the filter covers 2y-1 .. 2y+1;
    // bottom/right pad (pad_br): plane rows (y, y+1) hold image rows 2y .. 2y+3, the filter covers 2y .. 2y+2
    const int dy = pad_br ? (a == 0 ? py : (py == 0 ? 2 : -1)) : (a == 0 ? (py == 1 ? 0 : -1) : (py == 0 ? 1 : 2));
    const int dx = pad_br ? (b == 0 ? px : (px == 0 ? 2 : -1)) : (b == 0 ? (px == 1 ? 0 : -1) : (px == 0 ? 1 : 2));
    out[i] = (n < cout && dy >= 0 && dx >= 0) ? (T)w[((long)n * cin + c) * 9 + dy * 3 + dx] : (T)0.f;
  }
}

extern "C" int stedm_pack_conv_weight_s2d_frag(const float* w, void* out, int cout, int cin, int mm_dtype, int pad_br, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 8 == 0, "pack_conv_weight_s2d_frag: bad args (cin %% 8)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_s2d_frag: bad mm_dtype %d", mm_dtype);
  const long total = (long)((cout + 127) / 128) * (4 * cin / 16) * 4 * 4 * 64 * 8;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (mm_dtype == STEDM_F16) pack_conv_weight_s2d_frag_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)out, cout, cin, total, pad_br);
  else pack_conv_weight_s2d_frag_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)out, cout, cin, total, pad_br);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// The same two derived filters for the 3-product modes on the 16x16x32 MFMA kind (conv_rs.inc RS_SUBM): fragment order of
// stedm_pack_conv_weight_frag16 with 4 taps — out[parity][tn][chunk32][tap][c][lane][e] = W'[n = tn*128 + 16 c + (lane & 15)][ci = 32 chunk +
// 16 (g & 1) + 8 (g >> 1) + e][tap], g = lane >> 4 — as a hi stream followed by a lo stream (value - hi, rounded).
// UP: the four parity filters with pre-summed taps (pack_conv_weight_up_frag_kernel); else: the space-to-depth filter over 4 * cin channels.
template <typename T, bool UP>
__global__ void pack_conv_weight_sub_frag16_hl_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, long per_parity, long total,
                                                      int pad_br) {
  const int nch = (UP ? cin : 4 * cin) / 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int par = UP ? (int)(i / per_parity) : 0;
    const long ii = i - par * per_parity;
    const int e = (int)(ii & 7);
    const int lane = (int)((ii >> 3) & 63);
    const int c = (int)((ii >> 9) & 7);
    long r = ii >> 12;
    const int tap = (int)(r & 3); r >>= 2;
    const int chunk = (int)(r % nch);
    const int tn = (int)(r / nch);
    const int g = lane >> 4;
    const int n = tn * 128 + c * 16 + (lane & 15);
    const int cc = chunk * 32 + 16 * (g & 1) + 8 * (g >> 1) + e;
    const int a = tap >> 1, b = tap & 1;
    float v = 0.f;
    if (n < cout) {
      if (UP) {
        const int py = par >> 1, px = par & 1;
        const int dy0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), dy1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
        const int dx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), dx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
        for (int dy = dy0; dy <= dy1; ++dy)
          for (int dx = dx0; dx <= dx1; ++dx) v += w[((long)n * cin + cc) * 9 + dy * 3 + dx];
      } else {
        const int pb = cc / cin, ci = cc - pb * cin;          // parity block of the space-to-depth planes, channel inside it
        const int py = pb >> 1, px = pb & 1;
        const int dy = pad_br ? (a == 0 ? py : (py == 0 ? 2 : -1)) : (a == 0 ? (py == 1 ? 0 : -1) : (py == 0 ? 1 : 2));
        const int dx = pad_br ? (b == 0 ? px : (px == 0 ? 2 : -1)) : (b == 0 ? (px == 1 ? 0 : -1) : (px == 0 ? 1 : 2));
        if (dy >= 0 && dx >= 0) v = w[((long)n * cin + ci) * 9 + dy * 3 + dx];
      }
    }
    const T hv = (T)v;
    out[i] = hv;
    out[total + i] = (T)(v - (float)hv);
  }
}

extern "C" int stedm_pack_conv_weight_up_frag16_hl(const float* w, void* out, int cout, int cin, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 32 == 0 && cout > 0, "pack_conv_weight_up_frag16_hl: bad args (cin %% 32)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_up_frag16_hl: bad mm_dtype %d", mm_dtype);
  const long per = (long)((cout + 127) / 128) * (cin / 32) * 4 * 8 * 64 * 8, total = 4 * per;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (mm_dtype == STEDM_F16) pack_conv_weight_sub_frag16_hl_kernel<_Float16, true><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)out, cout, cin, per, total, 0);
  else pack_conv_weight_sub_frag16_hl_kernel<__bf16, true><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)out, cout, cin, per, total, 0);
  STEDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int stedm_pack_conv_weight_s2d_frag16_hl(const float* w, void* out, int cout, int cin, int mm_dtype, int pad_br, void* stream) {
  STEDM_CHECK_ARG(w && out && cin % 8 == 0 && cout > 0, "pack_conv_weight_s2d_frag16_hl: bad args (cin %% 8)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "pack_conv_weight_s2d_frag16_hl: bad mm_dtype %d", mm_dtype);
  const long total = (long)((cout + 127) / 128) * (4 * cin / 32) * 4 * 8 * 64 * 8;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (mm_dtype == STEDM_F16) pack_conv_weight_sub_frag16_hl_kernel<_Float16, false><<<grid, 256, 0, as_stream(stream)>>>(w, (_Float16*)out, cout, cin, total, total, pad_br);
  else pack_conv_weight_sub_frag16_hl_kernel<__bf16, false><<<grid, 256, 0, as_stream(stream)>>>(w, (__bf16*)out, cout, cin, total, total, pad_br);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// NHWC fp32 -> space-to-depth 16-bit planes [B][H/2][W/2][4C]; thread = one channel quad of one input pixel
template <typename T>
__global__ void __launch_bounds__(256) space_to_depth16_kernel(const float* __restrict__ x, T* __restrict__ hi, T* __restrict__ lo, int C, int H, int W,
                                                               long total_q, unsigned* ovf) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  const int Q = C >> 2;
  unsigned bad = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_q; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long pix = i / Q;
    const int xx = (int)(pix % W); pix /= W;
    const int yy = (int)(pix % H);
    const long b = pix / H;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const long o = ((((b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * 4 + ((yy & 1) * 2 + (xx & 1))) * (long)C >> 2) + q;
    V4 h4;
    h4[0] = (T)v.x; h4[1] = (T)v.y; h4[2] = (T)v.z; h4[3] = (T)v.w;
    reinterpret_cast<V4*>(hi)[o] = h4;
    bad |= f16_over_v4<T>(h4);
    if (lo) {
      V4 l4;
      l4[0] = (T)(v.x - (float)h4[0]); l4[1] = (T)(v.y - (float)h4[1]); l4[2] = (T)(v.z - (float)h4[2]); l4[3] = (T)(v.w - (float)h4[3]);
      reinterpret_cast<V4*>(lo)[o] = l4;
    }
  }
  f16_guard_commit(ovf, bad, STEDM_F16G_S2D);
}

extern "C" int stedm_space_to_depth16(const float* x, int C, int B, int H, int W, void* out_hi, void* out_lo, int mm_dtype, void* stream) {
  STEDM_CHECK_ARG(x && out_hi && C > 0 && C % 4 == 0 && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "space_to_depth16: bad args (C %% 4, even H/W)");
  STEDM_CHECK_ARG(mm_dtype == STEDM_F16 || mm_dtype == STEDM_BF16, "space_to_depth16: bad mm_dtype %d", mm_dtype);
  const long total_q = (long)B * H * W * (C / 4);
  const int grid = (int)((total_q + 255) / 256 < 16384 ? (total_q + 255) / 256 : 16384);
  if (mm_dtype == STEDM_F16) space_to_depth16_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>(x, (_Float16*)out_hi, (_Float16*)out_lo, C, H, W, total_q, f16_guard_flag());
  else space_to_depth16_kernel<__bf16><<<grid, 256, 0, as_stream(stream)>>>(x, (__bf16*)out_hi, (__bf16*)out_lo, C, H, W, total_q, nullptr);
  STEDM_LAUNCH_CHECK();
  return 0;
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = by + j, c = bx + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = in[(long)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = bx + j, r = by + threadIdx.x;
    if (r < rows && c < cols) out[(long)c * rows + r] = tile[threadIdx.x][j];
  }
}

extern "C" int stedm_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  STEDM_CHECK_ARG(in && out && rows > 0 && cols > 0, "transpose: bad args");
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
  transpose_kernel<<<grid, block, 0, as_stream(stream)>>>(in, out, rows, cols);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Embedding path
// ------------------------------------------------------------------------------------------------
// out[b][n] = act_out( bias[n] + sum_k act_in(x[b][k]) * wt[k][n] ),  wt K-major ([K][N]).
// Block = 64 outputs x 4 K-slices (256 threads), up to 8 batch rows per block share each weight read;
// the K split keeps the dependent-load chain short (these GEMVs are latency-, not bandwidth-bound).
constexpr int LIN_ROWS = 8;
constexpr int LIN_KC = 1024;   // K chunk staged in LDS
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B, int K, int N,
                                                     int act_in, int act_out) {
  __shared__ float sx[LIN_ROWS * LIN_KC];
  __shared__ float part[4 * LIN_ROWS * 64];
  const int b0 = blockIdx.y * LIN_ROWS;
  const int nl = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  float acc[LIN_ROWS];
#pragma unroll
  for (int r = 0; r < LIN_ROWS; ++r) acc[r] = 0.f;
  for (int kc = 0; kc < K; kc += LIN_KC) {
    const int kn = min(LIN_KC, K - kc);
    __syncthreads();
    for (int i = threadIdx.x; i < LIN_ROWS * kn; i += 256) {
      const int r = i / kn, k = i - r * kn;
      float v = (b0 + r < B) ? x[(long)(b0 + r) * K + kc + k] : 0.f;
      sx[r * LIN_KC + k] = act_in == 1 ? silu_f(v) : (act_in == 2 ? fmaxf(v, 0.f) : v);
    }
    __syncthreads();
    if (n < N) {
      const int per = (kn + 3) / 4;
      const int k0 = ks * per, k1 = min(kn, k0 + per);
#pragma unroll 8      // (eight weight loads in flight per thread: one per iteration made the B = 64 embedding Linears 40 us each, all latency)
      for (int k = k0; k < k1; ++k) {
        const float w = wt[(long)(kc + k) * N + n];
#pragma unroll
        for (int r = 0; r < LIN_ROWS; ++r) acc[r] = fmaf(sx[r * LIN_KC + k], w, acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LIN_ROWS; ++r) part[(ks * LIN_ROWS + r) * 64 + nl] = acc[r];
  __syncthreads();
  if (ks == 0 && n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < LIN_ROWS; ++r) {
      if (b0 + r < B) {
        float v = bv + ((part[(0 * LIN_ROWS + r) * 64 + nl] + part[(1 * LIN_ROWS + r) * 64 + nl]) +
                        (part[(2 * LIN_ROWS + r) * 64 + nl] + part[(3 * LIN_ROWS + r) * 64 + nl]));
        out[(long)(b0 + r) * N + n] = act_out == 1 ? silu_f(v) : (act_out == 2 ? fmaxf(v, 0.f) : v);
      }
    }
  }
}

// Few-row form (B <= ROWS, N % 4 == 0): the GEMV is a pure weight stream, so the block keeps many 16-B loads in flight -
// 16 output quads x 16 K-slices, 8 independent loads per thread - and adds the 16 slice partials in a fixed order.
template <int ROWS>
__global__ void __launch_bounds__(256) linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B, int K, int N,
                                                          int act_in, int act_out) {
  extern __shared__ float lsm[];
  float* sx = lsm;                         // [ROWS][K]
  float* part = lsm + ROWS * K;            // [16][ROWS][64]
  x += (long)blockIdx.y * ROWS * K;        // blockIdx.y: block of ROWS rows
  out += (long)blockIdx.y * ROWS * N;
  B = min(ROWS, B - (int)blockIdx.y * ROWS);
  const int q = threadIdx.x & 15, ks = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + q * 4;
#pragma unroll 4      // (unconditional loads from a clamped row, several in flight: a load inside the bounds branch is waited for at its join)
  for (int i = threadIdx.x; i < ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    float v = x[(long)min(r, B - 1) * K + k];
    v = r < B ? v : 0.f;
    sx[i] = act_in == 1 ? silu_f(v) : (act_in == 2 ? fmaxf(v, 0.f) : v);
  }
  __syncthreads();
  float acc[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
  if (n < N) {
    const int per = (K + 15) / 16;
    const int k0 = ks * per, k1 = min(K, k0 + per);
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
      const float4 w = *reinterpret_cast<const float4*>(wt + (long)k * N + n);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const float xv = sx[r * K + k];
        acc[r][0] = fmaf(xv, w.x, acc[r][0]); acc[r][1] = fmaf(xv, w.y, acc[r][1]);
        acc[r][2] = fmaf(xv, w.z, acc[r][2]); acc[r][3] = fmaf(xv, w.w, acc[r][3]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(ks * ROWS + r) * 64 + q * 4 + j] = acc[r][j];
  __syncthreads();
  const int nl = threadIdx.x & 63, r = threadIdx.x >> 6;      // 64 outputs x up to 4 rows per pass
  for (int rr = r; rr < ROWS && rr < B; rr += 4) {
    const int nn = blockIdx.x * 64 + nl;
    if (nn < N) {
      float v = bias ? bias[nn] : 0.f;
      for (int s2 = 0; s2 < 16; ++s2) v += part[(s2 * ROWS + rr) * 64 + nl];   // fixed order
      out[(long)rr * N + nn] = act_out == 1 ? silu_f(v) : (act_out == 2 ? fmaxf(v, 0.f) : v);
    }
  }
}

static int launch_linear(const float* x, const float* wt, const float* bias, float* out, int B, int K, int N, int act_in,
                         int act_out, hipStream_t st) {
  if (N % 4 == 0 && B <= 2 && (size_t)(2 * K + 16 * 2 * 64) * sizeof(float) <= 64 * 1024) {
    const size_t lds = (size_t)(2 * K + 16 * 2 * 64) * sizeof(float);
    linear_rows_kernel<2><<<(N + 63) / 64, 256, lds, st>>>(x, wt, bias, out, B, K, N, act_in, act_out);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  // 3 rows and more: the same weight stream in blocks of 8 rows (blockIdx.y), x rows resident in LDS — while the weights are re-read by few
  // enough row blocks (measured on MI355X, tools/bench_linear.py: B = 64: 7.7 / 9.9 / 38.8 us for 128 -> 512, 512 -> 512, 512 -> 10368 against
  // 9.4 / 25.9 / 53.8 us on the tiled GEMM below; B = 256, N = 10368: 136 against 85 us)
  if (N % 4 == 0 && (long)((B + 7) / 8) * ((N + 63) / 64) <= 2048 && (size_t)(8 * K + 16 * 8 * 64) * sizeof(float) <= 64 * 1024) {
    const size_t lds = (size_t)(8 * K + 16 * 8 * 64) * sizeof(float);
    linear_rows_kernel<8><<<dim3((N + 63) / 64, (B + 7) / 8), 256, lds, st>>>(x, wt, bias, out, B, K, N, act_in, act_out);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  if (B > LIN_ROWS) {      // many rows and a wide output: the tiled GEMM (sgemm.hpp), x read once per column tile
    stedm::SgemmArgs g{x, (long)K, 0, wt, (long)N, 0, out, (long)N, B, N, K, 1.0f, 0.0f, K, nullptr, bias, act_in, act_out};
    stedm::sgemm_launch(g, 1, st);
    STEDM_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((N + 63) / 64, (B + LIN_ROWS - 1) / LIN_ROWS);
  linear_kernel<<<grid, 256, 0, st>>>(x, wt, bias, out, B, K, N, act_in, act_out);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// sinusoid [B][mc]: cos half first (util.py:166)
__global__ void sinusoid_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ te, int mc) {
  const int b = blockIdx.x, half = mc / 2;
  const float tv = (float)t[b];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float a = tv * freqs[i];
    te[(long)b * mc + i] = cosf(a);
    te[(long)b * mc + half + i] = sinf(a);
  }
  if ((mc & 1) && threadIdx.x == 0) te[(long)b * mc + mc - 1] = 0.f;
}

// emb [B][ted]; ws: caller-provided scratch of B*(mc+ted) floats (sinusoid + hidden layer).
extern "C" int stedm_time_embed(const int64_t* t, const float* freqs, const float* w0t, const float* b0, const float* w2t,
                                const float* b2, float* emb, float* ws, int B, int mc, int ted, void* stream) {
  STEDM_CHECK_ARG(t && freqs && w0t && b0 && w2t && b2 && emb && ws, "time_embed: null pointer");
  STEDM_CHECK_ARG(B > 0 && mc > 0 && ted > 0, "time_embed: bad sizes B=%d mc=%d ted=%d", B, mc, ted);
  float* te = ws;
  float* h1 = ws + (size_t)B * mc;
  hipStream_t st = as_stream(stream);
  sinusoid_kernel<<<B, 64, 0, st>>>(t, freqs, te, mc);
  STEDM_LAUNCH_CHECK();
  int rc = launch_linear(te, w0t, b0, h1, B, mc, ted, 0, 1, st);
  if (rc) return rc;
  return launch_linear(h1, w2t, b2, emb, B, ted, ted, 0, 0, st);
}

// generic small Linear with optional input / output activation (0 none, 1 SiLU, 2 ReLU): Agg_Linear agg_blocks.py:14-18
extern "C" int stedm_linear(const float* x, const float* wt, const float* bias, float* out, int B, int k, int n, int act_in,
                            int act_out, void* stream) {
  STEDM_CHECK_ARG(x && wt && out && B > 0 && k > 0 && n > 0, "linear: bad args");
  return launch_linear(x, wt, bias, out, B, k, n, act_in, act_out, as_stream(stream));
}

extern "C" int stedm_emb_proj(const float* emb, const float* wt, const float* bias, float* out, int B, int k, int ntot,
                              void* stream) {
  STEDM_CHECK_ARG(emb && wt && bias && out, "emb_proj: null pointer");
  STEDM_CHECK_ARG(B > 0 && k > 0 && ntot > 0, "emb_proj: bad sizes B=%d k=%d ntot=%d", B, k, ntot);
  return launch_linear(emb, wt, bias, out, B, k, ntot, 1, 0, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// DDIM update + rescaled CFG. One block per sample; thread = (w, part) with part striding (c,h).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ e_c,
                                                        const float* __restrict__ e_u, const float* __restrict__ noise,
                                                        const float* __restrict__ coefs, const int32_t* __restrict__ step_idx,
                                                        float s, float phi, float* __restrict__ x_prev,
                                                        float* __restrict__ pred_x0, int C, int H, int W) {
  __shared__ float red[2][256];
  __shared__ float ratio_w[256];
  const int b = blockIdx.x;
  const int idx = step_idx ? *step_idx : 0;
  const float a_t = coefs[idx * 4 + 0], a_prev = coefs[idx * 4 + 1], sigma = coefs[idx * 4 + 2], sq1m = coefs[idx * 4 + 3];
  const int CH = C * H;
  const long base = (long)b * CH * W;
  const int parts = 256 / W;  // W <= 256 checked on host
  const int w = threadIdx.x % W, part = threadIdx.x / W;
  const bool active = part < parts;

  if (e_u) {
    // pass A: means over (c,h) for column w
    float sc = 0.f, sw = 0.f;
    if (active)
      for (int r = part; r < CH; r += parts) {
        const float ec = e_c[base + (long)r * W + w], eu = e_u[base + (long)r * W + w];
        sc += ec;
        sw += eu + s * (ec - eu);
      }
    red[0][threadIdx.x] = active ? sc : 0.f;
    red[1][threadIdx.x] = active ? sw : 0.f;
    __syncthreads();
    float mc = 0.f, mw = 0.f;
    for (int p = 0; p < parts; ++p) {
      mc += red[0][p * W + w];
      mw += red[1][p * W + w];
    }
    mc /= (float)CH;
    mw /= (float)CH;
    __syncthreads();
    // pass B: centred sums of squares -> unbiased std (torch.std default, ddim.py:183)
    float qc = 0.f, qw = 0.f;
    if (active)
      for (int r = part; r < CH; r += parts) {
        const float ec = e_c[base + (long)r * W + w], eu = e_u[base + (long)r * W + w];
        const float ew = eu + s * (ec - eu);
        qc += (ec - mc) * (ec - mc);
        qw += (ew - mw) * (ew - mw);
      }
    red[0][threadIdx.x] = active ? qc : 0.f;
    red[1][threadIdx.x] = active ? qw : 0.f;
    __syncthreads();
    if (threadIdx.x < W) {
      float vc = 0.f, vw = 0.f;
      for (int p = 0; p < parts; ++p) {
        vc += red[0][p * W + threadIdx.x];
        vw += red[1][p * W + threadIdx.x];
      }
      ratio_w[threadIdx.x] = sqrtf(vc / (float)(CH - 1)) / sqrtf(vw / (float)(CH - 1));
    }
    __syncthreads();
  }
  const float sqrt_at = sqrtf(a_t);
  const float dir_c = sqrtf(1.0f - a_prev - sigma * sigma);
  const float sqrt_ap = sqrtf(a_prev);
  if (active)
    for (int r = part; r < CH; r += parts) {
      const long o = base + (long)r * W + w;
      float e = e_c[o];
      if (e_u) {
        const float eu = e_u[o];
        const float ew = eu + s * (e - eu);
        e = (ew * ratio_w[w]) * phi + (1.0f - phi) * e;
      }
      const float x0 = (x[o] - sq1m * e) / sqrt_at;
      float xp = sqrt_ap * x0 + dir_c * e;
      if (noise) xp += sigma * noise[o];
      x_prev[o] = xp;
      if (pred_x0) pred_x0[o] = x0;
    }
}

// The same update with a thread's R = C H / (256 / W) elements of every operand held in registers: one round of independent loads instead of
// three dependent passes over e_c / e_u (the kernel above is one block per sample, i.e. 64 busy CUs at the bench batch: all latency, 24 us per
// step). Same operations in the same order as above (the sums run over r = part, part + parts, ...), so the results are the same bits.
template <int R>
__global__ void __launch_bounds__(256) ddim_step_reg_kernel(const float* __restrict__ x, const float* __restrict__ e_c, const float* __restrict__ e_u,
                                                            const float* __restrict__ noise, const float* __restrict__ coefs,
                                                            const int32_t* __restrict__ step_idx, float s, float phi, float* __restrict__ x_prev,
                                                            float* __restrict__ pred_x0, int C, int H, int W) {
  __shared__ float red[2][256];
  __shared__ float ratio_w[256];
  const int b = blockIdx.x;
  const int idx = step_idx ? *step_idx : 0;
  const float a_t = coefs[idx * 4 + 0], a_prev = coefs[idx * 4 + 1], sigma = coefs[idx * 4 + 2], sq1m = coefs[idx * 4 + 3];
  const int CH = C * H;
  const long base = (long)b * CH * W;
  const int parts = 256 / W;               // host: 256 % W == 0 and CH == R * parts, every thread active
  const int w = threadIdx.x % W, part = threadIdx.x / W;
  float ec[R], eu[R], xv[R], nz[R];
#pragma unroll
  for (int j = 0; j < R; ++j) ec[j] = e_c[base + (long)(part + j * parts) * W + w];
#pragma unroll
  for (int j = 0; j < R; ++j) xv[j] = x[base + (long)(part + j * parts) * W + w];
  if (e_u) {
#pragma unroll
    for (int j = 0; j < R; ++j) eu[j] = e_u[base + (long)(part + j * parts) * W + w];
  }
  if (noise) {
#pragma unroll
    for (int j = 0; j < R; ++j) nz[j] = noise[base + (long)(part + j * parts) * W + w];
  }
  if (e_u) {
    float sc = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      sc += ec[j];
      sw += eu[j] + s * (ec[j] - eu[j]);
    }
    red[0][threadIdx.x] = sc;
    red[1][threadIdx.x] = sw;
    __syncthreads();
    float mc = 0.f, mw = 0.f;
    for (int p = 0; p < parts; ++p) {
      mc += red[0][p * W + w];
      mw += red[1][p * W + w];
    }
    mc /= (float)CH;
    mw /= (float)CH;
    __syncthreads();
    float qc = 0.f, qw = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float ew = eu[j] + s * (ec[j] - eu[j]);
      qc += (ec[j] - mc) * (ec[j] - mc);
      qw += (ew - mw) * (ew - mw);
    }
    red[0][threadIdx.x] = qc;
    red[1][threadIdx.x] = qw;
    __syncthreads();
    if (threadIdx.x < W) {
      float vc = 0.f, vw = 0.f;
      for (int p = 0; p < parts; ++p) {
        vc += red[0][p * W + threadIdx.x];
        vw += red[1][p * W + threadIdx.x];
      }
      ratio_w[threadIdx.x] = sqrtf(vc / (float)(CH - 1)) / sqrtf(vw / (float)(CH - 1));
    }
    __syncthreads();
  }
  const float sqrt_at = sqrtf(a_t);
  const float dir_c = sqrtf(1.0f - a_prev - sigma * sigma);
  const float sqrt_ap = sqrtf(a_prev);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const long o = base + (long)(part + j * parts) * W + w;
    float e = ec[j];
    if (e_u) {
      const float ew = eu[j] + s * (e - eu[j]);
      e = (ew * ratio_w[w]) * phi + (1.0f - phi) * e;
    }
    const float x0 = (xv[j] - sq1m * e) / sqrt_at;
    float xp = sqrt_ap * x0 + dir_c * e;
    if (noise) xp += sigma * nz[j];
    x_prev[o] = xp;
    if (pred_x0) pred_x0[o] = x0;
  }
}

extern "C" int stedm_ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coefs,
                               const int32_t* step_idx, float cfg_scale, float rescale_phi, float* x_prev, float* pred_x0,
                               int B, int C, int H, int W, void* stream) {
  STEDM_CHECK_ARG(x && e_c && coefs && x_prev, "ddim_step: null pointer");
  STEDM_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && W <= 256, "ddim_step: bad shape B=%d C=%d H=%d W=%d (W <= 256)", B, C, H, W);
  STEDM_CHECK_ARG(!e_u || C * H > 1, "ddim_step: std over (C,H) needs C*H > 1");
  const int parts = 256 / W;
  const bool reg = 256 % W == 0 && (C * H) % parts == 0;
  const int per = reg ? C * H / parts : 0;
#define DDIM_REG(RR) ddim_step_reg_kernel<RR><<<B, 256, 0, as_stream(stream)>>>(x, e_c, e_u, noise, coefs, step_idx, cfg_scale, rescale_phi, x_prev, pred_x0, C, H, W)
  if (per == 16) DDIM_REG(16);          // 32 x 32 x 4 latents (the bench's)
  else if (per == 4) DDIM_REG(4);       // 16 x 16 x 4
  else if (per == 8) DDIM_REG(8);
  else
    ddim_step_kernel<<<B, 256, 0, as_stream(stream)>>>(x, e_c, e_u, noise, coefs, step_idx, cfg_scale, rescale_phi, x_prev,
                                                       pred_x0, C, H, W);
#undef DDIM_REG
  STEDM_LAUNCH_CHECK();
  return 0;
}

__global__ void step_advance_kernel(int32_t* p, int d) { *p += d; }
extern "C" int stedm_step_advance(int32_t* step_idx, int delta, void* stream) {
  STEDM_CHECK_ARG(step_idx, "step_advance: null pointer");
  step_advance_kernel<<<1, 1, 0, as_stream(stream)>>>(step_idx, delta);
  STEDM_LAUNCH_CHECK();
  return 0;
}

__global__ void step_set_t_kernel(const int64_t* __restrict__ ts, const int32_t* __restrict__ idx, int64_t* __restrict__ t, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) t[i] = ts[*idx];
}
extern "C" int stedm_step_set_t(const int64_t* ts_table, const int32_t* step_idx, int64_t* t_buf, int B, void* stream) {
  STEDM_CHECK_ARG(ts_table && step_idx && t_buf && B > 0, "step_set_t: bad args");
  step_set_t_kernel<<<(B + 255) / 256, 256, 0, as_stream(stream)>>>(ts_table, step_idx, t_buf, B);
  STEDM_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Graph helpers
// ------------------------------------------------------------------------------------------------
extern "C" int stedm_graph_begin(void* stream) {
  STEDM_HIP_TRY(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return 0;
}
extern "C" int stedm_graph_end(void* stream, void** graph_exec_out) {
  STEDM_CHECK_ARG(graph_exec_out, "graph_end: null out pointer");
  hipGraph_t g = nullptr;
  STEDM_HIP_TRY(hipStreamEndCapture(as_stream(stream), &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return 2;
  }
  *graph_exec_out = (void*)ge;
  return 0;
}
extern "C" int stedm_graph_launch(void* graph_exec, void* stream) {
  STEDM_CHECK_ARG(graph_exec, "graph_launch: null graph");
  STEDM_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, as_stream(stream)));
  return 0;
}
extern "C" int stedm_graph_destroy(void* graph_exec) {
  if (graph_exec) STEDM_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}


// ------------------------------------------------------------------------------------------------ per-sample normal noise
// x_T (ddim.py:122) and the per-step noise of eta > 0 (ddim.py:206) for a rank's shard of a data-parallel prediction run: row i depends only
// on (seed, stream, global sample id), so a sample is the same under any world size (the reference draws batch-shaped from the global
// generator, which cannot give that). Definition (the parity tests restate it in numpy): group g of four consecutive elements of a
// row = Philox4x32-10(counter {g, stream, 0x4E524D4C, 0}, key {seed, sample id}); words (w0, w1) and (w2, w3) give two Box-Muller pairs with
// u = (w + 0.5) 2^-32: z0 = sqrt(-2 ln u0) cos(2 pi u1), z1 = sqrt(-2 ln u0) sin(2 pi u1). fp32 arithmetic with the hardware log / sin / cos.
__global__ void __launch_bounds__(256) philox_normal_kernel(float* __restrict__ out, const long* __restrict__ ids, int id0, int n, unsigned seed, unsigned stream) {
  const int row = blockIdx.y;
  const unsigned sid = ids ? (unsigned)ids[row] : (unsigned)(id0 + row);
  const int ngroups = (n + 3) >> 2;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
    const U4 r = philox4x32_10(U4{(unsigned)g, stream, 0x4E524D4Cu, 0u}, seed, sid);
    float z[4];
    const float k = 2.3283064365386963e-10f;     // 2^-32
    const float u0 = ((float)r.x + 0.5f) * k, u1 = ((float)r.y + 0.5f) * k, u2 = ((float)r.z + 0.5f) * k, u3 = ((float)r.w + 0.5f) * k;
    const float ra = sqrtf(-2.0f * __logf(fminf(fmaxf(u0, 1.1641532e-10f), 0.99999994f))), rb = sqrtf(-2.0f * __logf(fminf(fmaxf(u2, 1.1641532e-10f), 0.99999994f)));
    z[0] = ra * __cosf(6.283185307179586f * u1); z[1] = ra * __sinf(6.283185307179586f * u1);
    z[2] = rb * __cosf(6.283185307179586f * u3); z[3] = rb * __sinf(6.283185307179586f * u3);
    float* o = out + (long)row * n + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * g + j < n) o[j] = z[j];
  }
}

extern "C" int stedm_philox_normal(float* out, int rows, int n, const long* sample_ids, int first_id, unsigned long long seed, unsigned stream, void* stream_) {
  STEDM_CHECK_ARG(out && rows > 0 && n > 0, "philox_normal: bad arguments");
  const int ngroups = (n + 3) / 4;
  dim3 grid((ngroups + 255) / 256 < 64 ? (ngroups + 255) / 256 : 64, rows);
  philox_normal_kernel<<<grid, 256, 0, as_stream(stream_)>>>(out, sample_ids, first_id, n, (unsigned)(seed & 0xFFFFFFFFull), stream);
  STEDM_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------ row-major im2col (16-bit), generic-shape fallback
// The implicit-GEMM kernels tile the output grid in runs of 128 / 256 pixels that must align with image rows (conv_geometry): latent widths
// that are not powers of two (96 x 96, 40 x 40, 24 x 24: anything UNetModel of the reference accepts, openaimodel.py:761-806 only needs
// H, W divisible by 4) have no such tiling. For those a 3x3 convolution runs as im2col + flat GEMM: dst [B][Ho][Wo][9 C] with column
// tap * C + c = src [b][sy][sx][c] (0 outside), then the 1x1 kind over K = 9 C with the ordinary [cout][tap][cin] weight planes (same
// contraction order as the tiled kernels' planes; every epilogue of the 1x1 kind applies). 9x the activation bytes: a correctness path.
// mode 0: stride 1 pad 1; 1: stride 2 pad 1 (Downsample.op, openaimodel.py:164-166); 2: nearest x2 then stride 1 pad 1 (Upsample, :129-131).
__global__ void __launch_bounds__(256) im2col_rows16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int Hs, int Ws, int C8, int Ho, int Wo, int mode) {
  const long total = (long)B * Ho * Wo * 9 * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    long r = i / C8;
    const int tap = (int)(r % 9); r /= 9;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    int sy, sx;
    bool ok;
    if (mode == 1) { sy = 2 * y + dy; sx = 2 * x + dx; ok = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws; }
    else if (mode == 2) { const int uy = y + dy, ux = x + dx; ok = uy >= 0 && uy < Ho && ux >= 0 && ux < Wo; sy = uy >> 1; sx = ux >> 1; }
    else { sy = y + dy; sx = x + dx; ok = sy >= 0 && sy < Hs && sx >= 0 && sx < Ws; }
    dst[i] = ok ? src[(((long)b * Hs + sy) * Ws + sx) * C8 + c8] : make_uint4(0u, 0u, 0u, 0u);
  }
}

extern "C" int stedm_im2col_rows16(const void* src16, void* dst16, int B, int Hs, int Ws, int C, int mode, void* stream) {
  STEDM_CHECK_ARG(src16 && dst16 && B > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0 && mode >= 0 && mode <= 2, "im2col_rows16: bad arguments (C %% 8 == 0)");
  const int Ho = mode == 1 ? (Hs - 1) / 2 + 1 : (mode == 2 ? 2 * Hs : Hs), Wo = mode == 1 ? (Ws - 1) / 2 + 1 : (mode == 2 ? 2 * Ws : Ws);
  const long total = (long)B * Ho * Wo * 9 * (C / 8);
  const long blocks = (total + 255) / 256;
  im2col_rows16_kernel<<<(int)(blocks < 65536 ? blocks : 65536), 256, 0, as_stream(stream)>>>((const uint4*)src16, (uint4*)dst16, B, Hs, Ws, C / 8, Ho, Wo, mode);
  STEDM_LAUNCH_CHECK();
  return 0;
}
