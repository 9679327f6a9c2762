// Cost of side instructions issued between the MFMAs of a wave that owns its SIMD (4 waves per CU, 256 CUs):
// time per MFMA slot when every slot (or every 4th slot) also issues one side instruction.  Results of reads are
// consumed 16 slots later so that only issue / pipeline cost is measured, not latency.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_side_cost mfma_side_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// KIND: 0 none, 1 ds_write_b128, 2 ds_read_b128, 3 global dwordx4 coalesced (1 KB per wave), 4 global dwordx4 in
// 32-byte pieces 512 B apart (32 cache lines per wave), 5 ds_write_b64, 6 ds_read_b64 stride 32 B (8-way conflict),
// 7 v_pk_add_f32 x4, 8 v_add_f32 x4, 9 ds_read_b64 contiguous, 10 global dwordx4 coalesced with a uniform base + 32-bit
// lane offset, 11 global -> LDS direct dwordx4 (no VGPR destination), 12 the same in 32-byte pieces
template <int KIND, int EVERY>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const int t = threadIdx.x;
  f32x4 ring[16];
  for (int i = 0; i < 16; ++i) ring[i] = f32x4{a, b, a, b};
  f32x4 sum = {0, 0, 0, 0};
  f32x2 pv[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  const f32x4* gco = (const f32x4*)in + t + (blockIdx.x & 63) * 4096;
  const f32x4* gsc = (const f32x4*)in + (t >> 1) * 32 + (t & 1) + (blockIdx.x & 63) * 4096;
  float* lw = lds + t * 4;
  const unsigned voff = t * 16;
  float* lr8 = lds + (t & 63) * 8;
  float* lr2 = lds + (t & 63) * 2;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x16* cc = (u & 3) == 0 ? &c0 : (u & 3) == 1 ? &c1 : (u & 3) == 2 ? &c2 : &c3;
      *cc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, *cc, 0, 0, 0);
      if ((u % EVERY) == 0) {
        if (KIND == 1) *(f32x4*)(lw + 1024 * (u & 3)) = ring[u];
        if (KIND == 2) { sum += ring[u]; ring[u] = *(volatile f32x4*)(lw + 1024 * (u & 3)); }
        if (KIND == 3) { sum += ring[u]; ring[u] = __builtin_nontemporal_load(gco + 256 * ((i * 16 + u) & 15)); }
        if (KIND == 4) { sum += ring[u]; ring[u] = __builtin_nontemporal_load(gsc + 2 * ((i * 16 + u) & 7)); }
        if (KIND == 5) *(f32x2*)(lw + 1024 * (u & 3)) = f32x2{ring[u][0], ring[u][1]};
        if (KIND == 6) { sum[0] += ring[u][0]; f32x2 r = *(volatile f32x2*)(lr8 + 1024 * (u & 3)); ring[u][0] = r[0] + r[1]; }
        if (KIND == 9) { sum[0] += ring[u][0]; f32x2 r = *(volatile f32x2*)(lr2 + 1024 * (u & 3)); ring[u][0] = r[0] + r[1]; }
        if (KIND == 10) { sum += ring[u]; ring[u] = *(const f32x4*)((const char*)in + (blockIdx.x & 63) * 65536 + 4096 * ((i * 16 + u) & 15) + voff); }
        if (KIND == 11) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gco + 256 * ((i * 16 + u) & 15)),
                                                         (__attribute__((address_space(3))) void*)(lds + 1024 * (u & 3) + 256 * (t >> 6)), 16, 0, 0);
        if (KIND == 12) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsc + 2 * ((i * 16 + u) & 7)),
                                                         (__attribute__((address_space(3))) void*)(lds + 1024 * (u & 3) + 256 * (t >> 6)), 16, 0, 0);
        if (KIND == 7) {
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pv[q]) : "v"(pv[q]), "v"(pv[(q + 1) & 3]));
        }
        if (KIND == 8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("v_add_f32 %0, %1, %2" : "=v"(pv[q][0]) : "v"(pv[q][0]), "v"(pv[(q + 1) & 3][1]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int i = 0; i < 16; ++i) sum += ring[i];
  float s = sum[0] + sum[1] + sum[2] + sum[3] + pv[0][0] + pv[1][1] + pv[2][0] + pv[3][1];
  for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
  if (s == 12345.678f) out[t] = s + lds[t];
}
template <int KIND, int EVERY>
void run(float* out, const float* in, int cus, const char* name, int wps = 1) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  k<KIND, EVERY><<<cus * wps, 256>>>(out, in, 1000, 1.0f, 0.5f);
  (void)hipEventRecord(e0);
  k<KIND, EVERY><<<cus * wps, 256>>>(out, in, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.37e9 / (iters * 16.0 * wps);
  printf("waves/SIMD %d  %-44s every %d MFMA: %6.1f cycles per MFMA slot  -> %6.1f extra cycles per side instruction\n", wps, name, EVERY, cyc,
         (cyc - 64.3) * EVERY);
}
int main() {
  float *out, *in; (void)hipMalloc(&out, 4096); (void)hipMalloc(&in, 8 << 20); (void)hipMemset(in, 0, 8 << 20);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int nb = p.multiProcessorCount;
  run<0, 1>(out, in, nb, "none");
  run<1, 1>(out, in, nb, "ds_write_b128");
  run<1, 4>(out, in, nb, "ds_write_b128");
  run<5, 1>(out, in, nb, "ds_write_b64");
  run<2, 1>(out, in, nb, "ds_read_b128");
  run<2, 4>(out, in, nb, "ds_read_b128");
  run<9, 1>(out, in, nb, "ds_read_b64 contiguous");
  run<6, 1>(out, in, nb, "ds_read_b64 32 B stride (8-way conflict)");
  run<6, 4>(out, in, nb, "ds_read_b64 32 B stride (8-way conflict)");
  run<3, 1>(out, in, nb, "global_load_dwordx4 coalesced");
  run<3, 4>(out, in, nb, "global_load_dwordx4 coalesced");
  run<4, 1>(out, in, nb, "global_load_dwordx4 32 B pieces");
  run<4, 4>(out, in, nb, "global_load_dwordx4 32 B pieces");
  run<10, 1>(out, in, nb, "global_load_dwordx4 coalesced, saddr + voffset");
  run<10, 4>(out, in, nb, "global_load_dwordx4 coalesced, saddr + voffset");
  run<11, 1>(out, in, nb, "global_load_lds_dwordx4 coalesced");
  run<11, 4>(out, in, nb, "global_load_lds_dwordx4 coalesced");
  run<12, 1>(out, in, nb, "global_load_lds_dwordx4 32 B pieces");
  run<12, 4>(out, in, nb, "global_load_lds_dwordx4 32 B pieces");
  run<0, 1>(out, in, nb, "none", 2);
  run<3, 1>(out, in, nb, "global_load_dwordx4 coalesced", 2);
  run<3, 4>(out, in, nb, "global_load_dwordx4 coalesced", 2);
  run<4, 4>(out, in, nb, "global_load_dwordx4 32 B pieces", 2);
  run<6, 1>(out, in, nb, "ds_read_b64 32 B stride (8-way conflict)", 2);
  run<9, 1>(out, in, nb, "ds_read_b64 contiguous", 2);
  run<2, 1>(out, in, nb, "ds_read_b128", 2);
  run<8, 1>(out, in, nb, "4 x v_add_f32", 2);
  run<7, 1>(out, in, nb, "4 x v_pk_add_f32", 2);
  run<7, 1>(out, in, nb, "4 x v_pk_add_f32");
  run<8, 1>(out, in, nb, "4 x v_add_f32");
  return 0;
}
