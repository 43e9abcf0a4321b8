// BASELINE config[4]: fp8 (OCP e4m3) MFMA attention for the ViT blocks -- Q K^T and P V on v_mfma_f32_32x32x16_fp8_fp8
// with f32 accumulation and f32 softmax, replacing vit_attn_kernel at the reference's call site video_attention.py:52.
//
// Two kernels behind maavss_vit_attn_fp8:
//   vit_qkv_fp8_kernel   one workgroup per (frame, head): absmax of that pair's q / k / v (16-bit, from the qkv GEMM), scales
//                        s = absmax / 448 (e4m3 max), q8 / k8 = x / s row-major [rows][heads*64] bytes, and V TRANSPOSED,
//                        vt8 [frame][head][64 d][ntok_pad] with the 16 keys of every MFMA k-step stored in k-slot order
//                        (position 8h + 4a + b holds key 8a + 4h + b), so that the P V A-operand is one 8-byte LDS read.
//   vit_attn_fp8_kernel  the flash loop of vit_attn.hip with 8-bit operands: K / V^T tiles are 4 KiB each (half the LDS
//                        bytes and fragment traffic), S = (K8 Q8^T) * sq*sk, P' = 2^7 exp2(s - m) <= 2^8 rounded to e4m3
//                        (the 2^7 keeps probabilities down to 2^-16 of the row maximum above e4m3's subnormal floor; it
//                        cancels in O / l because l sums the same scaled values), deferred rescale threshold 1 (log2).
// Query on the lane as in the 16-bit kernel: the exponentiated accumulator registers 8s..8s+7 of key block kb are the B
// fragment of k-step 2 kb + s.
#include <type_traits>
#include "mma.h"

#define F8_D 64
#define F8_QT 128
#define F8_KT 64
#define F8_THR 1.0f
#define F8_PSHIFT 7.0f
#define F8_MAX 448.0f

typedef __attribute__((ext_vector_type(2))) int i32x2_t;

__device__ __forceinline__ unsigned cvt_pk_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);   // bytes 0,1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);        // bytes 2,3
  return (unsigned)w;
}
__device__ __forceinline__ void mfma_fp8(f32x16& acc, uint2 a, uint2 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), acc, 0, 0, 0);
}

// ws layout (bytes): scales [frames*heads][4] f32 | q8 [rows][dim] | k8 [rows][dim] | vt8 [frames*heads][64][ntok_pad]
template <int MODE>
__global__ __launch_bounds__(256) void vit_qkv_fp8_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ scales,
                                                          unsigned char* __restrict__ q8, unsigned char* __restrict__ k8,
                                                          unsigned char* __restrict__ vt8, int ntok, int ntok_pad, int ld_qkv, int dim) {
  __shared__ float red[3][4];
  __shared__ __attribute__((aligned(16))) unsigned char vt[F8_D][F8_KT + 16];   // one transposed 64-key tile (+16: bank spread)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, head = blockIdx.x, frame = blockIdx.y, heads = gridDim.x;
  const int64_t row0 = (int64_t)frame * ntok;
  const int rsub = tid >> 3, c8 = (tid & 7) * 8;     // 32 rows per sweep, 8 elements (16 B) per thread
  const bf16_t* base = qkv + row0 * ld_qkv + head * F8_D + c8;
  float amax[3] = {0.f, 0.f, 0.f};
  for (int r = rsub; r < ntok; r += 32)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const uint4 u = *reinterpret_cast<const uint4*>(base + (int64_t)r * ld_qkv + m * dim);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        amax[m] = fmaxf(amax[m], fmaxf(fabsf(up16<MODE>((unsigned short)(w[e] & 0xffffu))), fabsf(up16<MODE>((unsigned short)(w[e] >> 16)))));
    }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const float v = wave_max(amax[m]);
    if (lane == 0) red[m][wv] = v;
  }
  __syncthreads();
  float sc[3], inv[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const float a = fmaxf(fmaxf(red[m][0], red[m][1]), fmaxf(red[m][2], red[m][3]));
    sc[m] = a > 0.f ? a * (1.f / F8_MAX) : 1.f;
    inv[m] = 1.f / sc[m];
  }
  const int group = frame * heads + head;
  if (tid < 3) scales[group * 4 + tid] = sc[tid];
  unsigned char* vtg = vt8 + (int64_t)group * F8_D * ntok_pad;
  for (int t0 = 0; t0 < ntok_pad; t0 += F8_KT) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kl = half * 32 + rsub, r = t0 + kl;            // key within the tile / token row
      float f[3][8];
      const bool ok = r < ntok;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (ok) u = *reinterpret_cast<const uint4*>(base + (int64_t)r * ld_qkv + m * dim);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[m][2 * e] = up16<MODE>((unsigned short)(w[e] & 0xffffu)) * inv[m];
          f[m][2 * e + 1] = up16<MODE>((unsigned short)(w[e] >> 16)) * inv[m];
        }
      }
      if (ok) {
        const uint2 qo = make_uint2(cvt_pk_fp8x4(f[0][0], f[0][1], f[0][2], f[0][3]), cvt_pk_fp8x4(f[0][4], f[0][5], f[0][6], f[0][7]));
        const uint2 ko = make_uint2(cvt_pk_fp8x4(f[1][0], f[1][1], f[1][2], f[1][3]), cvt_pk_fp8x4(f[1][4], f[1][5], f[1][6], f[1][7]));
        *reinterpret_cast<uint2*>(q8 + (row0 + r) * dim + head * F8_D + c8) = qo;
        *reinterpret_cast<uint2*>(k8 + (row0 + r) * dim + head * F8_D + c8) = ko;
      }
      // V: key kl = 16 step + 8 a + 4 h + b goes to position 16 step + 8 h + 4 a + b of its 8 d rows (zeros past the end):
      // four consecutive keys (same a, h) are four consecutive positions = one dword per d row.  The four lanes that hold
      // those keys for the same 8 d (lanes 8 apart) exchange their packed bytes and each writes the dwords of two d rows
      // (dword instead of byte LDS stores; the kernel's 0.52 ms per launch is memory latency -- one workgroup walks a
      // (frame, head)'s 785 rows twice with three loads in flight per thread -- not this exchange: DESIGN.md 9).
      const unsigned v01 = cvt_pk_fp8x4(f[2][0], f[2][1], f[2][2], f[2][3]), v23 = cvt_pk_fp8x4(f[2][4], f[2][5], f[2][6], f[2][7]);
      const int jq = rsub & 3, lbase = lane - 8 * jq;
      unsigned d01 = 0, d23 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned a01 = __shfl(v01, lbase + 8 * i, 64), a23 = __shfl(v23, lbase + 8 * i, 64);
        d01 |= ((a01 >> (8 * jq)) & 0xffu) << (8 * i);
        d23 |= ((a23 >> (8 * jq)) & 0xffu) << (8 * i);
      }
      const int kl0 = kl & ~3;
      const int pos0 = (kl0 & ~15) | ((kl0 & 4) << 1) | ((kl0 & 8) >> 1);
      *reinterpret_cast<unsigned*>(&vt[c8 + jq][pos0]) = d01;
      *reinterpret_cast<unsigned*>(&vt[c8 + 4 + jq][pos0]) = d23;
    }
    __syncthreads();
    {   // 64 d rows x 64 B: thread -> (d = tid >> 2, 16-B chunk tid & 3)
      const int d = tid >> 2, ch = (tid & 3) * 16;
      *reinterpret_cast<uint4*>(vtg + (int64_t)d * ntok_pad + t0 + ch) = *reinterpret_cast<const uint4*>(&vt[d][ch]);
    }
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void vit_attn_fp8_kernel(
    const float* __restrict__ scales, const unsigned char* __restrict__ q8, const unsigned char* __restrict__ k8,
    const unsigned char* __restrict__ vt8, bf16_t* __restrict__ out, int ntok, int ntok_pad, int ld_out, int dim, int heads,
    int qblocks, int ngroups) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[2][F8_KT * F8_D];   // [key][64 d] bytes, 8-B chunk ^ ((key>>2)&7)
  __shared__ __attribute__((aligned(16))) unsigned char Vs[2][F8_D * F8_KT];   // [d][64 key slots] bytes, 8-B chunk ^ ((d>>2)&7)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / qblocks) * 8 + xcd;
  if (group >= ngroups) return;
  const int qb = slot % qblocks;
  const int head = group % heads, frame = group / heads;
  const int q0 = qb * F8_QT + wv * 32;
  const bool wave_active = q0 < ntok;
  const int64_t row0 = (int64_t)frame * ntok;
  const float sq = scales[group * 4 + 0], sk = scales[group * 4 + 1], sv = scales[group * 4 + 2];
  const float sqk = sq * sk;

  uint2 fq[4];
  {
    int qr = q0 + r;
    qr = qr < ntok ? qr : ntok - 1;
    const unsigned char* qp = q8 + (row0 + qr) * dim + head * F8_D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const uint2*>(qp + 16 * ks);
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float mrow = 0.f, lrow = 0.f;

  // staging: one 16-B load per thread and matrix.  K tile: thread -> (key = tid >> 2, 16-B chunk tid & 3) of k8 rows;
  // V^T tile: thread -> (d = tid >> 2, chunk tid & 3) of the pair's vt8 rows.  Keys past the end: K rows clamped to the
  // last token (their scores are masked), V^T columns are zero padding.
  const unsigned char* kframe = k8 + row0 * dim + head * F8_D + (tid & 3) * 16;
  const unsigned char* vframe = vt8 + ((int64_t)group * F8_D + (tid >> 2)) * ntok_pad + (tid & 3) * 16;
  const int skey = tid >> 2;
  uint4 kreg, vreg;
  int ktile0 = 0;
  auto load_tile = [&]() {
    int key = ktile0 + skey;
    key = key < ntok ? key : ntok - 1;
    kreg = *reinterpret_cast<const uint4*>(kframe + (int64_t)key * dim);
    vreg = *reinterpret_cast<const uint4*>(vframe + ktile0);
    ktile0 += F8_KT;
  };
  auto store_tile = [&](int buf) {
    const int row = tid >> 2, c2 = (tid & 3) * 2, x = (row >> 2) & 7;     // two 8-B chunks c2, c2+1 of a 64-B row
    const uint2 k0 = make_uint2(kreg.x, kreg.y), k1 = make_uint2(kreg.z, kreg.w);
    const uint2 v0 = make_uint2(vreg.x, vreg.y), v1 = make_uint2(vreg.z, vreg.w);
    *reinterpret_cast<uint2*>(&Ks[buf][row * 64 + ((c2 ^ x) * 8)]) = k0;
    *reinterpret_cast<uint2*>(&Ks[buf][row * 64 + (((c2 + 1) ^ x) * 8)]) = k1;
    *reinterpret_cast<uint2*>(&Vs[buf][row * 64 + ((c2 ^ x) * 8)]) = v0;
    *reinterpret_cast<uint2*>(&Vs[buf][row * 64 + (((c2 + 1) ^ x) * 8)]) = v1;
  };
  int koff[4], voff[4];   // byte offsets: K fragment of d-step ks (key r), V^T fragment of k-step (d row r)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    koff[ks] = r * 64 + (((2 * ks + h) ^ ((r >> 2) & 7)) * 8);
    voff[ks] = koff[ks];   // same shape: row = d within the 32-row block, chunk = 2 * kstep + h
  }

  const int ntiles = (ntok + F8_KT - 1) / F8_KT;
  const bool last_half = ntok - (ntiles - 1) * F8_KT <= 32;
  load_tile();
  store_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  auto max3 = [](float a, float b, float c) __attribute__((always_inline)) {
    float d;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  };
  auto tile = [&](int kt, auto last_c, auto nkb_c) __attribute__((always_inline)) {
    constexpr int NKB = decltype(nkb_c)::value;
    const int buf = kt & 1, kv0 = kt * F8_KT;
    const unsigned char* kt_base = &Ks[buf][0];
    const unsigned char* vt_base = &Vs[buf][0];
    __builtin_amdgcn_s_setprio(1);
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint2 fk = *reinterpret_cast<const uint2*>(kt_base + kb * 32 * 64 + koff[ks]);
        mfma_fp8(s[kb], fk, fq[ks]);
      }
    }
    // dequantise and subtract the running maximum (the 16-bit kernel folds -m into the accumulator's initial value; here the
    // product carries sq*sk, so it is one fused multiply-add per score): s' = acc * sq*sk - (m - 7)
    const float mshift = F8_PSHIFT - mrow;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = fmaf(s[kb][e], sqk, mshift);
    if (decltype(last_c)::value && kv0 + NKB * 32 > ntok) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= ntok) s[kb][e] = -1e30f;
    }
    __builtin_amdgcn_s_setprio(0);
    float mx;
    {
      float a[NKB][5];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int g = 0; g < 5; ++g) a[kb][g] = max3(s[kb][3 * g], s[kb][3 * g + 1], s[kb][3 * g + 2]);
        a[kb][0] = max3(a[kb][0], a[kb][1], s[kb][15]);
        a[kb][2] = max3(a[kb][2], a[kb][3], a[kb][4]);
      }
      if constexpr (NKB == 2) mx = max3(max3(a[0][0], a[0][2], a[1][0]), a[1][2], a[1][2]);
      else mx = max3(a[0][0], a[0][2], a[0][2]);
    }
    mx -= F8_PSHIFT;                       // back to "score minus running maximum"
    const bool first = kt == 0;
    if (__any(first || mx > F8_THR)) {
      float ma, mb;
      lane_swap32(mx, ma, mb);
      mx = fmaxf(ma, mb);
      const float delta = (first || mx > F8_THR) ? mx : 0.f;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      mrow += delta;
      lrow *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kb][e] -= delta;
    }
    uint2 fp[NKB][2];
    {
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(s[kb][e]);
          s[kb][e] = p;
          sum += p;
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
          fp[kb][sx] = make_uint2(cvt_pk_fp8x4(s[kb][8 * sx], s[kb][8 * sx + 1], s[kb][8 * sx + 2], s[kb][8 * sx + 3]),
                                  cvt_pk_fp8x4(s[kb][8 * sx + 4], s[kb][8 * sx + 5], s[kb][8 * sx + 6], s[kb][8 * sx + 7]));
      }
      lrow += sum;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const uint2 fv = *reinterpret_cast<const uint2*>(vt_base + db * 32 * 64 + voff[2 * kb + sx]);
          mfma_fp8(o[db], fv, fp[kb][sx]);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int kt = 0; kt < ntiles - 1; ++kt) {
    load_tile();
    __builtin_amdgcn_sched_barrier(0);
    if (wave_active) tile(kt, std::false_type{}, std::integral_constant<int, 2>{});
    asm volatile("" : "+v"(kreg.x), "+v"(kreg.y), "+v"(kreg.z), "+v"(kreg.w));
    asm volatile("" : "+v"(vreg.x), "+v"(vreg.y), "+v"(vreg.z), "+v"(vreg.w));
    store_tile((kt & 1) ^ 1);
    __syncthreads();
  }
  if (wave_active) {
    if (last_half) tile(ntiles - 1, std::true_type{}, std::integral_constant<int, 1>{});
    else tile(ntiles - 1, std::true_type{}, std::integral_constant<int, 2>{});
  }
  {
    float la, lb;
    lane_swap32(lrow, la, lb);
    const float inv = sv / (la + lb);
    const int qr = q0 + r;
    if (qr < ntok) {
      bf16_t* op = out + (row0 + qr) * ld_out + head * F8_D + 4 * h;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = pack2<MODE>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          u.y = pack2<MODE>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + 32 * db + 8 * g) = u;
        }
    }
  }
}

static int64_t fp8_ntok_pad(int ntok) { return ((int64_t)ntok + F8_KT - 1) / F8_KT * F8_KT; }

extern "C" int64_t maavss_vit_attn_fp8_ws_bytes(int frames, int ntok, int heads) {
  const int64_t groups = (int64_t)frames * heads, rows = (int64_t)frames * ntok, dim = (int64_t)heads * F8_D;
  return groups * 16 + 2 * rows * dim + groups * F8_D * fp8_ntok_pad(ntok) + 256;
}

extern "C" int maavss_vit_attn_fp8(const void* qkv, void* out, void* ws, int frames, int ntok, int heads, int ld_qkv, int ld_out,
                                   int dtype, void* stream) {
  MAAVSS_CHECK_ARG(qkv && out && ws && frames > 0 && ntok > 0, "vit_attn_fp8: bad arguments");
  MAAVSS_CHECK_ARG(heads >= 1 && ld_qkv >= 3 * heads * F8_D && ld_out >= heads * F8_D && ld_qkv % 8 == 0 && ld_out % 4 == 0,
                   "vit_attn_fp8: bad layout");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_attn_fp8: dtype (of qkv / out) must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(((uintptr_t)ws & 15) == 0, "vit_attn_fp8: ws must be 16-byte aligned");
  const int dim = heads * F8_D;
  const int64_t groups = (int64_t)frames * heads, rows = (int64_t)frames * ntok;
  const int ntok_pad = (int)fp8_ntok_pad(ntok);
  float* scales = (float*)ws;
  unsigned char* q8 = (unsigned char*)ws + ((groups * 16 + 255) / 256) * 256;
  unsigned char* k8 = q8 + rows * dim;
  unsigned char* vt8 = k8 + rows * dim;
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* q = (const bf16_t*)qkv;
  if (dtype == MODE_F16)
    hipLaunchKernelGGL(vit_qkv_fp8_kernel<MODE_F16>, dim3(heads, frames), dim3(256), 0, st, q, scales, q8, k8, vt8, ntok, ntok_pad, ld_qkv, dim);
  else
    hipLaunchKernelGGL(vit_qkv_fp8_kernel<MODE_BF16>, dim3(heads, frames), dim3(256), 0, st, q, scales, q8, k8, vt8, ntok, ntok_pad, ld_qkv, dim);
  MAAVSS_LAUNCH_CHECK("vit_qkv_fp8_kernel");
  const int qblocks = cdiv(ntok, F8_QT), ngroups = frames * heads;
  const int nblocks = cdiv(ngroups, 8) * 8 * qblocks;
  if (dtype == MODE_F16)
    hipLaunchKernelGGL(vit_attn_fp8_kernel<MODE_F16>, dim3(nblocks), dim3(256), 0, st, scales, q8, k8, vt8, (bf16_t*)out, ntok, ntok_pad,
                       ld_out, dim, heads, qblocks, ngroups);
  else
    hipLaunchKernelGGL(vit_attn_fp8_kernel<MODE_BF16>, dim3(nblocks), dim3(256), 0, st, scales, q8, k8, vt8, (bf16_t*)out, ntok, ntok_pad,
                       ld_out, dim, heads, qblocks, ngroups);
  MAAVSS_LAUNCH_CHECK("vit_attn_fp8_kernel");
  return MAAVSS_OK;
}
