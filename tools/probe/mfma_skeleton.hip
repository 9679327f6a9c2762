// Skeleton of one chunk of the stride-2 Winograd kernels (winograd_s2.hip): 72 MFMAs on 18 accumulators (16 in AGPRs,
// 2 in VGPRs), 27 ds_read_b128 fragment reads, one barrier -- what does the skeleton alone cost per chunk?
// VAR bit 0: barrier per chunk, bit 1: fragment reads issued, bit 2: MFMA operands come from the fragments (waits),
// bit 3: 8 MFMAs of a position run mi-major (4 + 4) instead of alternating, bit 4: 4 ds_write_b64 per position on top
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_skeleton mfma_skeleton.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mfma_a(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
constexpr int SLABV = 516, SLABU = 260;
template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave & 1, wn = wave >> 1, li = lane & 31, lh = lane >> 5;
  for (int i = t; i < 18 * SLABV + 18 * SLABU; i += 256) lds[i] = a0 + (i & 7) * b0;
  __syncthreads();
  const float* vb_ = lds + lh * SLABV + (wm * 64 + li) * 4;
  const float* ub_ = lds + 18 * SLABV + lh * SLABU + (wn * 32 + li) * 4;
  float* wr_ = lds + 18 * SLABV + 18 * SLABU + t * 2;
  f32x16 acc[9][2];
  for (int xi = 0; xi < 9; ++xi) for (int mi = 0; mi < 2; ++mi) for (int r = 0; r < 16; ++r) acc[xi][mi][r] = 0.f;
  f32x4 fa_[2][2], fb_[2];
  fa_[0][0] = fa_[0][1] = fa_[1][0] = fa_[1][1] = fb_[0] = fb_[1] = f32x4{a0, b0, a0, b0};
  f32x2 wv = {a0, b0};
  for (int it = 0; it < iters; ++it) {
    if (VAR & 2) {
      fa_[0][0] = *(const f32x4*)(vb_);
      fa_[0][1] = *(const f32x4*)(vb_ + 128);
      fb_[0] = *(const f32x4*)(ub_);
    }
#define SLOT(SIDX)                                                                                  \
    {                                                                                               \
      constexpr int sidx = (SIDX);                                                                  \
      constexpr int xi = sidx >> 3, w_ = sidx & 7;                                                  \
      constexpr int mi = (VAR & 8) ? (w_ >> 2) : (w_ & 1), e = (VAR & 8) ? (w_ & 3) : (w_ >> 1);    \
      if ((VAR & 4) && w_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | ((xi == 0 ? 0 : ((VAR & 16) ? 4 : 0)) << 8));  \
      if (xi < 8) mfma_a(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                          \
      else mfma_v(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                                 \
      if ((VAR & 2) && w_ < 3 && xi + 1 < 9) {                                                      \
        constexpr int nx = (xi + 1 < 9) ? xi + 1 : 0;                                               \
        if (w_ == 0) fa_[nx & 1][0] = *(const f32x4*)(vb_ + nx * 2 * SLABV);                        \
        if (w_ == 1) fa_[nx & 1][1] = *(const f32x4*)(vb_ + nx * 2 * SLABV + 128);                  \
        if (w_ == 2) fb_[nx & 1] = *(const f32x4*)(ub_ + nx * 2 * SLABU);                           \
      }                                                                                             \
      if ((VAR & 16) && w_ >= 4) *(f32x2*)(wr_ + (w_ - 4) * 512) = wv;                              \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define SLOT8(B) SLOT((B)) SLOT((B) + 1) SLOT((B) + 2) SLOT((B) + 3) SLOT((B) + 4) SLOT((B) + 5) SLOT((B) + 6) SLOT((B) + 7)
    SLOT8(0) SLOT8(8) SLOT8(16) SLOT8(24) SLOT8(32) SLOT8(40) SLOT8(48) SLOT8(56) SLOT8(64)
    if (VAR & 1) __syncthreads();
  }
  float s = 0.f;
  for (int xi = 0; xi < 9; ++xi) for (int mi = 0; mi < 2; ++mi) for (int r = 0; r < 16; ++r) s += acc[xi][mi][r];
  if (s == 12345.678f) out[t] = s;
}
template <int VAR>
void run(float* out, int nb, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  const size_t ldsb = (18 * SLABV + 18 * SLABU + 4096) * 4;
  (void)hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  k<VAR><<<nb, 256, ldsb>>>(out, 200, 1.0f, 0.37f);
  (void)hipEventRecord(e0);
  k<VAR><<<nb, 256, ldsb>>>(out, iters, 1.0f, 0.37f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("VAR %2d  %-58s %7.1f us/1000 chunks -> %6.0f cycles/chunk at 2.3 GHz (MFMA only: 4608)  %.1f TF\n", VAR, name,
         ms * 1e3 / iters * 1000, ms * 1e-3 / iters * 2.3e9, nb * 4.0 * 72 * 4096 * iters / (ms * 1e-3) * 1e-12);
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int nb = p.multiProcessorCount;
  run<0>(out, nb, "MFMAs only, alternating accumulators");
  run<8>(out, nb, "MFMAs only, 4 + 4 per accumulator");
  run<1>(out, nb, "+ barrier");
  run<3>(out, nb, "+ barrier + fragment reads (unused)");
  run<7>(out, nb, "+ barrier + fragment reads feeding the MFMAs");
  run<15>(out, nb, "same, 4 + 4");
  run<23>(out, nb, "+ 4 ds_write_b64 per position");
  return 0;
}
