// K4/K5: multi-head self-attention of the ViT blocks (softmax(Q K^T / 8) V, 6 heads x 64) and the CLS-row
// attention of the last block -- what dino's Attention.forward / get_last_selfattention compute for the
// reference's video_attention.py:52-56.
//
// vit_attn_kernel: flash-style, never materialises the N x N scores.  Workgroup = 64 query rows of one
// (frame, head); each of the 4 waves owns 16 rows.  K/V tiles of 64 keys are staged in LDS (K XOR-swizzled for
// ds_read_b128, V row-major for ds_read_b64_tr_b16); S = Q K^T and O += P V run on v_mfma_f32_16x16x32_bf16;
// the online softmax (running max / sum per query row) uses 16-lane shuffles; P goes through a per-wave LDS
// tile to become the A operand of P V.  qkv is the fused projection output [rows][1152] (q already scaled by
// 1/8 in the GEMM epilogue), out is [rows][384] with heads concatenated, both bf16.
// vit_cls_attn_kernel: last block only -- the CLS query against all keys, softmax over N tokens, the CLS
// column dropped (video_attention.py:56): att [frames][6][N-1] f32.
#include "mma.h"

#define ATT_D 64
#define ATT_QT 64
#define ATT_KT 64
#define P_LD 72  // bf16 elements per P row in LDS (144 B: 16-B aligned)

__global__ __launch_bounds__(256) void vit_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int ntok,
                                                       int ld_qkv, int ld_out, int dim) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[ATT_KT * ATT_D];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[ATT_KT * ATT_D];
  __shared__ __attribute__((aligned(16))) bf16_t Ps[4 * 16 * P_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, gq = lane >> 4;
  const int q0 = blockIdx.x * ATT_QT, head = blockIdx.y, frame = blockIdx.z;
  const int64_t row0 = (int64_t)frame * ntok;
  const bf16_t* qbase = qkv + head * ATT_D;
  const bf16_t* kbase = qkv + dim + head * ATT_D;
  const bf16_t* vbase = qkv + 2 * dim + head * ATT_D;

  // Q fragments of this wave's 16 rows (rows past the end are clamped; their results are never stored)
  bf16x8 fq[2];
  {
    int qr = q0 + wv * 16 + l16;
    qr = qr < ntok ? qr : ntok - 1;
    const bf16_t* qp = qbase + (row0 + qr) * ld_qkv;
    fq[0] = *reinterpret_cast<const bf16x8*>(qp + gq * 8);
    fq[1] = *reinterpret_cast<const bf16x8*>(qp + 32 + gq * 8);
  }
  f32x4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[4], lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrow[r] = -1e30f; lrow[r] = 0.f; }
  bf16_t* pw = Ps + wv * 16 * P_LD;

  for (int kv0 = 0; kv0 < ntok; kv0 += ATT_KT) {
    __syncthreads();
    // stage K (swizzled) and V tiles: 64 keys x 128 B each, 16-B chunks
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = it * 256 + tid, key = q >> 3, c = q & 7;
      uint4 kvv = make_uint4(0, 0, 0, 0), vvv = make_uint4(0, 0, 0, 0);
      if (kv0 + key < ntok) {
        kvv = *reinterpret_cast<const uint4*>(kbase + (row0 + kv0 + key) * ld_qkv + c * 8);
        vvv = *reinterpret_cast<const uint4*>(vbase + (row0 + kv0 + key) * ld_qkv + c * 8);
      }
      *reinterpret_cast<uint4*>(Ks + key * ATT_D + ((c ^ (key & 7)) * 8)) = kvv;
      *reinterpret_cast<uint4*>(Vs + key * ATT_D + c * 8) = vvv;
    }
    __syncthreads();
    // S = Q K^T  (16 rows x 64 keys per wave)
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key = nt * 16 + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 fk = *reinterpret_cast<const bf16x8*>(Ks + key * ATT_D + (((ks * 4 + gq) ^ (key & 7)) * 8));
        Mma<MODE_BF16>::mma(s[nt], fq[ks], fk);
      }
      if (kv0 + key >= ntok) s[nt] = f32x4{-1e30f, -1e30f, -1e30f, -1e30f};
    }
    // online softmax: row (4*gq + r) lives in the 16 lanes that share gq
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float mn = fmaxf(mrow[r], mx);
      alpha[r] = __expf(mrow[r] - mn);
      mrow[r] = mn;
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float p = __expf(s[nt][r] - mn);
        s[nt][r] = p;
        sum += p;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off, 64);
      lrow[r] = lrow[r] * alpha[r] + sum;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[d][r] *= alpha[r];
    // P -> LDS (C layout -> A operand layout)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(gq * 4 + r) * P_LD + nt * 16 + l16] = f2bf(s[nt][r]);
    __syncthreads();
    // O += P V
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 fp = *reinterpret_cast<const bf16x8*>(pw + l16 * P_LD + ks * 32 + gq * 8);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        bf16x4 h[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int key = ks * 32 + gq * 8 + hh * 4 + (l16 >> 2);
          const bf16_t* a = Vs + key * ATT_D + d * 16 + (l16 & 3) * 4;
          h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
        }
        const bf16x8 fv = bf16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
        Mma<MODE_BF16>::mma(o[d], fp, fv);
      }
    }
  }
  // normalise and store
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qr = q0 + wv * 16 + gq * 4 + r;
    if (qr < ntok) {
      const float inv = 1.f / lrow[r];
      bf16_t* op = out + (row0 + qr) * ld_out + head * ATT_D;
#pragma unroll
      for (int d = 0; d < 4; ++d) op[d * 16 + l16] = f2bf(o[d][r] * inv);
    }
  }
}

// one block per (frame, head): scores of the CLS query against all tokens
__global__ __launch_bounds__(256) void vit_cls_attn_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ att, int ntok,
                                                           int ld_qkv, int dim) {
  extern __shared__ float sc[];  // [ntok]
  __shared__ float qv[ATT_D];
  __shared__ float red[4];
  const int tid = threadIdx.x, head = blockIdx.x, frame = blockIdx.y;
  const int64_t row0 = (int64_t)frame * ntok;
  if (tid < ATT_D) qv[tid] = bf2f(qkv[row0 * ld_qkv + head * ATT_D + tid]);
  __syncthreads();
  float mx = -1e30f;
  for (int j = tid; j < ntok; j += 256) {
    const bf16_t* kp = qkv + (row0 + j) * ld_qkv + dim + head * ATT_D;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 u = *reinterpret_cast<const uint4*>(kp + c * 8);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = fmaf(__uint_as_float(w[e] << 16), qv[c * 8 + 2 * e], acc);
        acc = fmaf(__uint_as_float(w[e] & 0xffff0000u), qv[c * 8 + 2 * e + 1], acc);
      }
    }
    sc[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < ntok; j += 256) {
    const float p = __expf(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  float* ap = att + ((int64_t)frame * gridDim.x + head) * (ntok - 1);
  for (int j = tid + 1; j < ntok; j += 256) ap[j - 1] = sc[j] * inv;
}

extern "C" int maavss_vit_attn(const void* qkv, void* out, int frames, int ntok, int heads, int ld_qkv, int ld_out,
                               void* stream) {
  MAAVSS_CHECK_ARG(qkv && out && frames > 0 && ntok > 0, "vit_attn: bad arguments");
  MAAVSS_CHECK_ARG(heads >= 1 && ld_qkv >= 3 * heads * ATT_D && ld_out >= heads * ATT_D && ld_qkv % 8 == 0, "vit_attn: bad layout");
  hipLaunchKernelGGL(vit_attn_kernel, dim3(cdiv(ntok, ATT_QT), heads, frames), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qkv, (bf16_t*)out, ntok, ld_qkv, ld_out, heads * ATT_D);
  MAAVSS_LAUNCH_CHECK("vit_attn_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_cls_attn(const void* qkv, float* att, int frames, int ntok, int heads, int ld_qkv, void* stream) {
  MAAVSS_CHECK_ARG(qkv && att && frames > 0 && ntok > 1, "vit_cls_attn: bad arguments");
  MAAVSS_CHECK_ARG((size_t)ntok * 4 <= 60 * 1024, "vit_cls_attn: too many tokens for the LDS score buffer");
  hipLaunchKernelGGL(vit_cls_attn_kernel, dim3(heads, frames), dim3(256), ntok * sizeof(float), (hipStream_t)stream,
                     (const bf16_t*)qkv, att, ntok, ld_qkv, heads * ATT_D);
  MAAVSS_LAUNCH_CHECK("vit_cls_attn_kernel");
  return MAAVSS_OK;
}
