// Instruction-issue probe for gfx950 (measurement tool, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 -o valu_probe.bin valu_probe.hip && ./valu_probe.bin
// Part 1: cycles per wave-instruction of the vector instructions the ViT epilogues are made of, one wave per SIMD and three waves
//         per SIMD (s_memtime around 64 x 16 independent instructions).
// Part 2: the fc1 question: a wave's loop is {24 dependent v_mfma_f32_32x32x16_f16, then NV vector instructions}; with
//         three such waves per SIMD, is the loop time max(MFMA, VALU) or their sum?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { I_FMA32, I_PKFMA32, I_PKMUL32, I_PKFMA16, I_PKMUL16, I_PKMAX16, I_MED3, I_EXP32, I_EXP16, I_CVTPK16, I_MAX3, I_ADD32, I_DOT2, I_PKADD16, I_MUL32, I_MAX32, I_FMAC32, I_FMA32_S, I_FMA32_3V, I_PKFMA16_3V, I_PKADD32, I_SUB32, I_MOV, I_MAX32_E64, I_CVTPKBF16, I_XOR, I_FMAAK, I_MUL32_E64, I_PKMUL16_3V, I_LSHL, I_PERM, I_CNDMASK, I_MAX3_3V, I_MAX32_2V, I_EXP32_D, I_CVTPK16_2V, I_N };
static const char* kNames[I_N] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_fma_f16", "v_pk_mul_f16", "v_pk_max_f16", "v_med3_f32",
                                  "v_exp_f32", "v_exp_f16", "v_cvt_pk_f16_f32", "v_max3_f32", "v_add_f32", "v_dot2c_f32_f16", "v_pk_add_f16", "v_mul_f32", "v_max_f32", "v_fmac_f32", "v_fma_f32(v,s,s)", "v_fma_f32(3 vgpr)", "v_pk_fma_f16(3 vgpr)", "v_pk_add_f32", "v_sub_f32", "v_mov_b32", "v_max_f32_e64", "v_cvt_pk_bf16_f32", "v_xor_b32", "v_fmaak_f32", "v_mul_f32_e64", "v_pk_mul_f16(2 vgpr)", "v_lshlrev_b32", "v_perm_b32", "v_cndmask_b32", "v_max3_f32(3 vgpr)", "v_max_f32(d!=s)", "v_exp_f32(d!=s)", "v_cvt_pk_f16(2 vgpr)"};

template <int I>
__device__ __forceinline__ void one(float& a, v2f& p, unsigned& h, float c, v2f pc, unsigned hc) {
  if constexpr (I == I_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(pc));
  if constexpr (I == I_PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc));
  if constexpr (I == I_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_PKMAX16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  if constexpr (I == I_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(h));
  if constexpr (I == I_CVTPK16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(h) : "v"(a));
  if constexpr (I == I_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_DOT2) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(a) : "v"(hc));
  if constexpr (I == I_PKADD16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_MAX32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_FMAC32) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_FMA32_S) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "s"(c));
  if constexpr (I == I_FMA32_3V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(pc.y));
  if constexpr (I == I_PKFMA16_3V) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(h) : "v"(hc), "v"(c));
  if constexpr (I == I_PKADD32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pc));
  if constexpr (I == I_SUB32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(c));
  if constexpr (I == I_MAX32_E64) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_CVTPKBF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(h) : "v"(a));
  if constexpr (I == I_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_FMAAK) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(a) : "v"(c));
  if constexpr (I == I_MUL32_E64) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a) : "v"(c));
  if constexpr (I == I_PKMUL16_3V) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h) : "v"(hc), "v"(c));
  if constexpr (I == I_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(h));
  if constexpr (I == I_PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(h) : "v"(hc));
  if constexpr (I == I_MAX3_3V) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(pc.y));
  if constexpr (I == I_MAX32_2V) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(c), "v"(pc.y));
  if constexpr (I == I_EXP32_D) asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(c));
  if constexpr (I == I_CVTPK16_2V) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(c), "v"(pc.y));
  if constexpr (I == I_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(h) : "v"(hc));
}

template <int I>
__global__ void rate_kernel(long long* cycles, float* sink, int reps) {
  float a[16];
  v2f p[16];
  unsigned h[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = 0.5f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] * 0.5f}; h[i] = 0x3c003800u + i; }
  const float c = 0.999f;
  const v2f pc = {0.999f, 1.001f};
  const unsigned hc = 0x3bff3c01u;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#define STEP(i) one<I>(a[i], p[i], h[i], c, pc, hc);
    REP16(STEP) REP16(STEP) REP16(STEP) REP16(STEP)
#undef STEP
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)h[i];
  if (s == 1234.567f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Part 2: {24 dependent MFMA ; NV x VALU of kind I} per iteration
template <int I, int NV, bool MFMA>
__global__ __launch_bounds__(768) void mix_kernel(long long* cycles, float* sink, int reps) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  h8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (threadIdx.x & 7) + 0.001f * i); fb[i] = (_Float16)(0.02f * i - 0.05f); }
  float a[16];
  v2f p[16];
  unsigned h[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = 0.5f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] * 0.5f}; h[i] = 0x3c003800u + i; }
  const float c = 0.999f;
  const v2f pc = {0.999f, 1.001f};
  const unsigned hc = 0x3bff3c01u;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if constexpr (MFMA) {
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) one<I>(a[k & 15], p[k & 15], h[k & 15], c, pc, hc);
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)h[i] + acc[i];
  if (s == 1234.567f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Part 3: the MFMA shape under the power-limited clock (MI355X_MICROARCH.md, DVFS give-back item 7): the same FLOPs as 32x32x16 or
// 16x16x32 instructions on RANDOM operands, register-resident, three waves per SIMD, timed with HIP events over many launches.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(768) void shape_kernel(const _Float16* src, float* sink, int iters) {
  h8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 8 + i) * 8) % 32768);
    b[i] = *reinterpret_cast<const h8*>(src + ((threadIdx.x * 8 + 4 + i) * 8 + 64 * blockIdx.x) % 32768);
  }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b[(k + 1) & 3], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 2) & 3], b[k & 3], acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  } else {
    f32x4v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + j) & 3], b[(k + (j >> 1)) & 3], acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  }
  if (s == 1234.567f) sink[0] = s;
}

template <int SHAPE>
static void run_shape(const _Float16* d_src, float* d_sink, const char* what) {
  const int iters = 4000, launches = 30;      // 16 (32x32x16) or 32 (16x16x32) MFMAs per iteration = 524288 FLOP per wave and iteration
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(shape_kernel<SHAPE>, dim3(256), dim3(768), 0, 0, d_src, d_sink, iters);
  hipEventRecord(e0, 0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(shape_kernel<SHAPE>, dim3(256), dim3(768), 0, 0, d_src, d_sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 524288.0 * iters * 12 * 256 * launches;
  printf("  %-12s %s operands: %.1f ms for %d launches, %.0f TFLOP/s\n", SHAPE == 32 ? "32x32x16" : "16x16x32", what, ms, launches, flop / (ms * 1e-3) / 1e12);
}

static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

template <int I>
static void run_rate(long long* d_cyc, float* d_sink) {
  for (int waves : {4, 12}) {
    const int reps = 64, blocks = 256;
    hipLaunchKernelGGL(rate_kernel<I>, dim3(blocks), dim3(waves * 64), 0, 0, d_cyc, d_sink, reps);
    hipLaunchKernelGGL(rate_kernel<I>, dim3(blocks), dim3(waves * 64), 0, 0, d_cyc, d_sink, reps);
    hipDeviceSynchronize();
    std::vector<long long> c(blocks * waves);
    hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const double per = median(c) / (reps * 64.0);
    printf("  %-18s %2d waves/CU: %6.2f cycles per instruction per wave  (= %5.2f per SIMD-instruction at %d waves/SIMD)\n", kNames[I], waves, per,
           per / (waves / 4), waves / 4);
  }
}

template <int I, int NV>
static void run_mix(long long* d_cyc, float* d_sink) {
  const int reps = 200, blocks = 256, waves = 12;
  double res[3];
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL((mix_kernel<I, 0, true>), dim3(blocks), dim3(waves * 64), 0, 0, d_cyc, d_sink, reps);
      if (mode == 1) hipLaunchKernelGGL((mix_kernel<I, NV, false>), dim3(blocks), dim3(waves * 64), 0, 0, d_cyc, d_sink, reps);
      if (mode == 2) hipLaunchKernelGGL((mix_kernel<I, NV, true>), dim3(blocks), dim3(waves * 64), 0, 0, d_cyc, d_sink, reps);
    }
    hipDeviceSynchronize();
    std::vector<long long> c(blocks * waves);
    hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    res[mode] = median(c) / reps;
  }
  printf("  24 MFMA + %3d x %-22s 3 waves/SIMD, cycles per iteration per wave: MFMA only %7.0f | VALU only %7.0f | both %7.0f  (max %7.0f, sum %7.0f)\n", NV,
         kNames[I], res[0], res[1], res[2], std::max(res[0], res[1]), res[0] + res[1]);
}

int main() {
  long long* d_cyc;
  float* d_sink;
  hipMalloc(&d_cyc, 256 * 16 * sizeof(long long));
  hipMalloc(&d_sink, 64);
  printf("# part 1: issue cost (s_memtime cycles; 64 x 64 independent instructions per wave)\n");
  run_rate<I_FMA32>(d_cyc, d_sink); run_rate<I_ADD32>(d_cyc, d_sink); run_rate<I_MUL32>(d_cyc, d_sink); run_rate<I_MAX32>(d_cyc, d_sink);
  run_rate<I_PKFMA32>(d_cyc, d_sink); run_rate<I_PKFMA16>(d_cyc, d_sink); run_rate<I_EXP32>(d_cyc, d_sink); run_rate<I_CVTPK16>(d_cyc, d_sink);
  run_rate<I_MOV>(d_cyc, d_sink); run_rate<I_FMAC32>(d_cyc, d_sink); run_rate<I_MAX3>(d_cyc, d_sink);
  printf("# part 2: {24 dependent v_mfma_f32_32x32x16_f16 ; NV vector instructions} per loop iteration, 12 waves per CU\n");
    run_mix<I_MAX3, 16>(d_cyc, d_sink); run_mix<I_MAX3_3V, 16>(d_cyc, d_sink); run_mix<I_MAX3_3V, 64>(d_cyc, d_sink); run_mix<I_MAX32_2V, 32>(d_cyc, d_sink); run_mix<I_MAX32_2V, 64>(d_cyc, d_sink);
  run_mix<I_EXP32_D, 32>(d_cyc, d_sink); run_mix<I_CVTPK16_2V, 16>(d_cyc, d_sink); run_mix<I_ADD32, 32>(d_cyc, d_sink);
  run_mix<I_FMA32, 64>(d_cyc, d_sink); run_mix<I_FMA32_S, 64>(d_cyc, d_sink); run_mix<I_FMA32_3V, 64>(d_cyc, d_sink); run_mix<I_FMAC32, 64>(d_cyc, d_sink);
  run_mix<I_FMAAK, 64>(d_cyc, d_sink); run_mix<I_MUL32, 64>(d_cyc, d_sink); run_mix<I_MUL32_E64, 64>(d_cyc, d_sink); run_mix<I_ADD32, 64>(d_cyc, d_sink);
  run_mix<I_SUB32, 64>(d_cyc, d_sink); run_mix<I_MAX32, 64>(d_cyc, d_sink); run_mix<I_MAX32_E64, 64>(d_cyc, d_sink); run_mix<I_MAX3, 64>(d_cyc, d_sink);
  run_mix<I_MOV, 64>(d_cyc, d_sink); run_mix<I_XOR, 64>(d_cyc, d_sink); run_mix<I_LSHL, 64>(d_cyc, d_sink); run_mix<I_PERM, 64>(d_cyc, d_sink);
  run_mix<I_CNDMASK, 64>(d_cyc, d_sink); run_mix<I_PKFMA16, 64>(d_cyc, d_sink); run_mix<I_PKFMA16_3V, 64>(d_cyc, d_sink); run_mix<I_PKMUL16_3V, 64>(d_cyc, d_sink);
  run_mix<I_PKADD32, 64>(d_cyc, d_sink); run_mix<I_PKMUL32, 64>(d_cyc, d_sink); run_mix<I_PKFMA32, 64>(d_cyc, d_sink);
  run_mix<I_CVTPK16, 64>(d_cyc, d_sink); run_mix<I_CVTPKBF16, 64>(d_cyc, d_sink); run_mix<I_EXP32, 32>(d_cyc, d_sink);
  printf("# part 3: MFMA shape vs held clock (f16, register operands, 3 waves per SIMD, 256 workgroups)\n");
  {
    std::vector<_Float16> h(32768);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 9) % 2001 - 1000) * 1e-3f); }
    _Float16* d_src;
    hipMalloc(&d_src, h.size() * 2);
    hipMemcpy(d_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { run_shape<32>(d_src, d_sink, "random"); run_shape<16>(d_src, d_sink, "random"); }
    hipMemset(d_src, 0, h.size() * 2);
    run_shape<32>(d_src, d_sink, "zero  "); run_shape<16>(d_src, d_sink, "zero  ");
  }
  return 0;
}
